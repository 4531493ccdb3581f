// orb_kernels.hpp -- device-side structs and kernel declarations of the ORB extractor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbfe.h"
#include "../../include/orbfe_math.h"

namespace orbfe {

// A batch of same-sized images: frame f starts at base + f * fstride, rows are `pitch` bytes apart.
struct ImgView {
    const uint8_t* base;
    uint8_t* base_w; // same memory, writable (null for read-only views)
    size_t fstride;
    int pitch;
};

// Geometry of one pyramid level, computed on the host with the reference's own float arithmetic
// (ORBextractor.cc:767-787, :546-558) and read by every kernel.
struct LevelGeom {
    int w, h;            // level image size (:1112)
    int pitch;           // row pitch of the level inside the pyramid block (levels >= 1)
    int bpitch;          // row pitch inside the blurred block
    long long img_off;   // byte offset of the level inside a frame's pyramid block (levels >= 1)
    long long blur_off;  // byte offset inside a frame's blurred block
    int nCols, nRows, wCell, hCell; // FAST cell grid (:784-787)
    int maxBX, maxBY;    // maxBorderX/Y (:774-775); minBorder is 16
    int cell_first;      // index of the level's first active cell in the frame's cell array
    int ncells;          // active cells (rows/cols skipped at :795,:804 are not materialised)
    int cell_cap;        // candidate slots per cell: ceil(wCell/2)*ceil(hCell/2) (no two NMS survivors are adjacent)
    long long slot_off;  // u32 offset of the level's slots inside a frame's slot block
    int cand_cap;        // ncells * cell_cap
    long long cand_off;  // u32 offset (per ping-pong half) inside a frame's key scratch
    int quota;           // mnFeaturesPerLevel[level] (:435-446)
    int out_cap;         // slots for the level's distributed keypoints
    int out_off;         // u32 offset inside a frame's lvl_out block
    int nIni;            // root nodes (:543)
    float hX;            // (:545)
    float scale;         // mvScaleFactor[level]
    float kp_size;       // (float)(int)(31 * scale) (:837)
};

// The hardware hands workgroups to the 8 XCDs round-robin by linear workgroup id, and every XCD has its own L2.
// Kernels whose workgroups of one frame read overlapping image regions are launched as a 1-D grid of
// xcd_grid(nx * nframes) workgroups and call xcd_remap() for (bx, f): consecutive logical ids -- a frame's nx
// workgroups -- then run on the same XCD and share its L2 instead of fetching the same lines into all eight.
inline int xcd_grid(int total) { return ((total + 7) / 8) * 8; }
#if defined(__HIPCC__)
__device__ __forceinline__ bool xcd_remap(int nx, int total, int& bx, int& f)
{
    const int wg = blockIdx.x, per = (total + 7) >> 3;
    const int logical = (wg & 7) * per + (wg >> 3);
    if (logical >= total) return false;
    f = logical / nx;
    bx = logical - f * nx;
    return true;
}
#endif

__global__ void k_resize_level(ImgView src, ImgView dst, int sw, int sh, int dw4, int dh, double scale_x,
                               double scale_y, int dw);
#define RS_ROWS 8
__global__ void k_resize_tab(ImgView src, ImgView dst, int sw, int sh, int dw4, int dh, int nthreads, const int* xofs,
                             const int* xal, const int4* ytab, int nx, int total);
__global__ void k_fast_cells(ImgView src0, ImgView pyr, const LevelGeom* geom, const uint32_t* cellinfo,
                             uint32_t* slots, size_t slots_fstride, int32_t* cellcnt, int ncells_total, int iniTh,
                             int minTh, int roi_pitch, int roi_rows, int map_pitch, int map_rows, int list_cap, int nx, int total,
                             int cell_base, int cell_end);
__global__ void k_distribute(const LevelGeom* geom, const uint32_t* slots, size_t slots_fstride,
                             const int32_t* cellcnt, int ncells_total, uint32_t* keyscratch, size_t keys_fstride,
                             uint32_t* lvl_out, int out_fstride, int32_t* lvl_cnt, int nlevels, int32_t* lvl_ncand,
                             int keycap_lds, int nodecap, int veccap, const int32_t* worklist, const int32_t* worklist_n);
#define QT_MAXROOTS 16   // root nodes of DistributeOctTree (nIni = round(width / height), ORBextractor.cc:544) the kernels hold
#define QP_THREADS 256
__global__ void k_distribute_pyr(const LevelGeom* geom, const uint32_t* slots, size_t slots_fstride,
                                 const int32_t* cellcnt, int ncells_total, uint32_t* lvl_out, int out_fstride,
                                 int32_t* lvl_cnt, int nlevels, int32_t* lvl_ncand, int32_t* fallback, int D,
                                 int nodecap, int veccap, int32_t* worklist, int32_t* worklist_n, int by_level);

inline size_t qp_lds_bytes(int nIni, int D, int nodecap, int veccap)
{
    const size_t nleaf = (size_t)nIni << (2 * D);
    const size_t T = (size_t)nIni * (((1u << (2 * (D + 1))) - 1) / 3);
    size_t b = nleaf * 4 + (size_t)veccap * 16 + (size_t)nodecap * 24;
    b += 2 * (((size_t)nodecap * 2 + 15) & ~(size_t)15);
    b += (T / 2 + 4) * 4;
    return b + 32;
}
__global__ void k_level_offsets(const int32_t* lvl_cnt, int32_t* lvl_off, int32_t* n_out, int nlevels, int nframes,
                                int capacity, int32_t* overflow, const LevelGeom* geom, const uint32_t* lvl_out,
                                int out_fstride, uint32_t* flat_kv, uint8_t* flat_lvl, int32_t* worklist_n);
// k_blur7_mfma: a 32-column strip of a level per wave; c0 / c1 / c2 = byte columns of the three 16-byte pieces of a row it loads, tab =
// index (units of 64 uint4) of the strip's two pass-1 tap matrices in operand layout (orb_extractor.hip: build_blur_tables)
struct BlurStrip { int level, x0, c0, c1, c2, tab; };
__global__ void k_blur7_mfma(ImgView src0, ImgView pyr, ImgView blur, const LevelGeom* geom, const BlurStrip* strips, const uint4* tabs,
                             const uint4* tab2, int K2, int nstrips, int nx, int total);
template <bool ED> __global__ void k_blur7(ImgView src0, ImgView pyr, ImgView blur, const LevelGeom* geom, const uint32_t* strips, int nx,
                        int total);
#ifndef OD2_LDS_PAD
#define OD2_LDS_PAD 8192   // k_orient_describe2: unused LDS that caps its workgroups at seven a CU (see the kernel)
#endif
__global__ void k_orient_describe2(ImgView src0, ImgView pyr, ImgView blur, const LevelGeom* geom,
                                  const uint32_t* flat_kv, const uint8_t* flat_lvl, const int32_t* n_out, int nlevels,
                                  const uint32_t* pattern32, const uint4* icw, orbfe_keypoint* kps, uint8_t* desc,
                                  int capacity, int nx, int total);
__global__ void k_unpack_keys(const uint32_t* in, int n, int add, orbfe_keypoint* out);

inline size_t qt_lds_bytes(int keycap_lds, int nodecap, int veccap)
{
    size_t b = (size_t)veccap * 16;                 // vec + vprev
    b += (size_t)nodecap * 12;                      // begin, count, seq
    b += (size_t)nodecap * 14;                      // x0,y0,x1,y1,next,prev,free
    b += ((size_t)nodecap + 15) & ~(size_t)15;      // flags
    b = (b + 15) & ~(size_t)15;
    b += (size_t)keycap_lds * 8;                    // two key buffers
    return b + 16;
}

} // namespace orbfe
