// On-disk keyframe feature records of the reference's map files (src/Map.cc:297-321 SaveKeyFrame, :478-511 LoadKeyFrame):
// per feature  pt.x pt.y size angle response (5 x f32) | octave (i32) | mDescriptors.cols (i32, always 32) | 32 descriptor
// bytes | map point index (u64, ULONG_MAX = none)  = 68 bytes = 17 dwords, packed back to back after the keyframe header.
// Every offset in the file is a multiple of 4, so the records are moved as dwords.  Pure byte shuffling, HBM-bound:
// a workgroup stages 256 records (17 KB) through LDS so that both the file side and the array side are read and written
// as contiguous dwords; the stride of 17 dwords is odd, i.e. free of LDS bank conflicts.
#include "orbfe_common.hpp"

using namespace orbfe;

#define KF_REC_DWORDS 17
#define KF_THREADS 256

namespace {

// file image -> arrays.  Segment s (one keyframe): records at byte offset seg_off[s] (multiple of 4), nrec = seg_first[s+1] -
// seg_first[s], first output index seg_first[s].  seg_off == NULL: one segment at offset 0 with n records.
__global__ __launch_bounds__(KF_THREADS) void k_kf_unpack(const uint32_t* __restrict__ file, const unsigned long long* __restrict__ seg_off,
                                                         const int32_t* __restrict__ seg_first, int n_single,
                                                         uint32_t* __restrict__ kps, uint32_t* __restrict__ desc,
                                                         uint32_t* __restrict__ mp, int32_t* __restrict__ bad)
{
    __shared__ uint32_t s_rec[KF_THREADS * KF_REC_DWORDS];
    const int s = blockIdx.y;
    const size_t src0 = seg_off ? (size_t)(seg_off[s] >> 2) : 0;
    const int first = seg_first ? seg_first[s] : 0;
    const int nrec = seg_first ? seg_first[s + 1] - first : n_single;
    const int r0 = blockIdx.x * KF_THREADS;
    if (r0 >= nrec) return;
    const int cnt = min(KF_THREADS, nrec - r0);
    const uint32_t* src = file + src0 + (size_t)r0 * KF_REC_DWORDS;
    {
        uint32_t v[KF_REC_DWORDS]; // all loads of the tile in flight before the first LDS store
#pragma unroll
        for (int k = 0; k < KF_REC_DWORDS; k++) {
            const int j = threadIdx.x + k * KF_THREADS;
            v[k] = j < cnt * KF_REC_DWORDS ? src[j] : 0u;
        }
#pragma unroll
        for (int k = 0; k < KF_REC_DWORDS; k++) {
            const int j = threadIdx.x + k * KF_THREADS;
            if (j < cnt * KF_REC_DWORDS) s_rec[j] = v[k];
        }
    }
    __syncthreads();
    const size_t o = (size_t)first + r0;
    for (int j = threadIdx.x; j < cnt * 7; j += KF_THREADS) { // cv::KeyPoint: 6 stored fields + class_id (not stored: -1)
        const int r = j / 7, c = j - r * 7;
        kps[o * 7 + j] = c < 6 ? s_rec[r * KF_REC_DWORDS + c] : 0xffffffffu;
    }
    for (int j = threadIdx.x; j < cnt * 8; j += KF_THREADS) desc[o * 8 + j] = s_rec[(j >> 3) * KF_REC_DWORDS + 7 + (j & 7)];
    if (mp)
        for (int j = threadIdx.x; j < cnt * 2; j += KF_THREADS) mp[o * 2 + j] = s_rec[(j >> 1) * KF_REC_DWORDS + 15 + (j & 1)];
    if (threadIdx.x < cnt && s_rec[threadIdx.x * KF_REC_DWORDS + 6] != 32u) atomicAdd(bad, 1); // Descriptors.cols (Map.cc:492)
}

__global__ __launch_bounds__(KF_THREADS) void k_kf_pack(const uint32_t* __restrict__ kps, const uint32_t* __restrict__ desc,
                                                       const uint32_t* __restrict__ mp, const unsigned long long* __restrict__ seg_off,
                                                       const int32_t* __restrict__ seg_first, int n_single, uint32_t* __restrict__ file)
{
    __shared__ uint32_t s_rec[KF_THREADS * KF_REC_DWORDS];
    const int s = blockIdx.y;
    const size_t dst0 = seg_off ? (size_t)(seg_off[s] >> 2) : 0;
    const int first = seg_first ? seg_first[s] : 0;
    const int nrec = seg_first ? seg_first[s + 1] - first : n_single;
    const int r0 = blockIdx.x * KF_THREADS;
    if (r0 >= nrec) return;
    const int cnt = min(KF_THREADS, nrec - r0);
    const size_t o = (size_t)first + r0;
    {
        uint32_t a[7], b[8], c2[2]; // the tile's 17 loads per thread in flight together
#pragma unroll
        for (int k = 0; k < 7; k++) { const int j = threadIdx.x + k * KF_THREADS; a[k] = j < cnt * 7 ? kps[o * 7 + j] : 0u; }
#pragma unroll
        for (int k = 0; k < 8; k++) { const int j = threadIdx.x + k * KF_THREADS; b[k] = j < cnt * 8 ? desc[o * 8 + j] : 0u; }
#pragma unroll
        for (int k = 0; k < 2; k++) { const int j = threadIdx.x + k * KF_THREADS; c2[k] = (mp && j < cnt * 2) ? mp[o * 2 + j] : 0xffffffffu; } // ULONG_MAX = no map point (Map.cc:316-317)
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const int j = threadIdx.x + k * KF_THREADS;
            const int r = j / 7, c = j - r * 7;
            if (j < cnt * 7 && c < 6) s_rec[r * KF_REC_DWORDS + c] = a[k];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int j = threadIdx.x + k * KF_THREADS;
            if (j < cnt * 8) s_rec[(j >> 3) * KF_REC_DWORDS + 7 + (j & 7)] = b[k];
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int j = threadIdx.x + k * KF_THREADS;
            if (j < cnt * 2) s_rec[(j >> 1) * KF_REC_DWORDS + 15 + (j & 1)] = c2[k];
        }
    }
    if (threadIdx.x < cnt) s_rec[threadIdx.x * KF_REC_DWORDS + 6] = 32u; // mDescriptors.cols
    __syncthreads();
    uint32_t* dst = file + dst0 + (size_t)r0 * KF_REC_DWORDS;
    for (int j = threadIdx.x; j < cnt * KF_REC_DWORDS; j += KF_THREADS) dst[j] = s_rec[j];
}

struct KfWorkspace {
    DevBuf file, kps, desc, mp, bad;
};
thread_local ThreadWorkspaces<KfWorkspace> tl_kf_ws; // per (thread, device): host-pointer entry points only
KfWorkspace& kf_ws() { return tl_kf_ws.get(); }

} // namespace

extern "C" {

int orbfe_keyframe_features_unpack_device(const uint8_t* d_file, const uint64_t* d_seg_offset, const int32_t* d_seg_first, int nsegments,
                                          int max_records_per_segment, orbfe_keypoint* d_kps, uint8_t* d_desc, uint64_t* d_mp_index,
                                          int32_t* d_bad, void* stream)
{
    if (nsegments < 0 || max_records_per_segment < 0 || (nsegments && max_records_per_segment && (!d_file || !d_kps || !d_desc || !d_bad)) ||
        ((d_seg_offset == nullptr) != (d_seg_first == nullptr)) || (!d_seg_offset && nsegments > 1) || ((uintptr_t)d_file & 3))
        return fail(ORBFE_ERR_INVALID, "orbfe_keyframe_features_unpack_device: invalid argument");
    if (nsegments == 0 || max_records_per_segment == 0) return ORBFE_OK;
    hipLaunchKernelGGL(k_kf_unpack, dim3((max_records_per_segment + KF_THREADS - 1) / KF_THREADS, nsegments), dim3(KF_THREADS), 0,
                       (hipStream_t)stream, (const uint32_t*)d_file, (const unsigned long long*)d_seg_offset, d_seg_first,
                       max_records_per_segment, (uint32_t*)d_kps, (uint32_t*)d_desc, (uint32_t*)d_mp_index, d_bad);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_keyframe_features_pack_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const uint64_t* d_mp_index,
                                        const uint64_t* d_seg_offset, const int32_t* d_seg_first, int nsegments,
                                        int max_records_per_segment, uint8_t* d_file, void* stream)
{
    if (nsegments < 0 || max_records_per_segment < 0 || (nsegments && max_records_per_segment && (!d_file || !d_kps || !d_desc)) ||
        ((d_seg_offset == nullptr) != (d_seg_first == nullptr)) || (!d_seg_offset && nsegments > 1) || ((uintptr_t)d_file & 3))
        return fail(ORBFE_ERR_INVALID, "orbfe_keyframe_features_pack_device: invalid argument");
    if (nsegments == 0 || max_records_per_segment == 0) return ORBFE_OK;
    hipLaunchKernelGGL(k_kf_pack, dim3((max_records_per_segment + KF_THREADS - 1) / KF_THREADS, nsegments), dim3(KF_THREADS), 0,
                       (hipStream_t)stream, (const uint32_t*)d_kps, (const uint32_t*)d_desc, (const uint32_t*)d_mp_index,
                       (const unsigned long long*)d_seg_offset, d_seg_first, max_records_per_segment, (uint32_t*)d_file);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_keyframe_features_pack(const orbfe_keypoint* kps, const uint8_t* desc, const uint64_t* mp_index, int n, uint8_t* out, int device)
{
    if (n < 0 || (n && (!kps || !desc || !out))) return fail(ORBFE_ERR_INVALID, "orbfe_keyframe_features_pack: invalid argument");
    int rc = use_device(device);
    if (rc || n == 0) return rc;
    KfWorkspace& w = kf_ws();
    const size_t N = (size_t)n;
    if ((rc = w.file.ensure(N * ORBFE_KF_FEATURE_BYTES)) || (rc = w.kps.ensure(N * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure(N * 32)) ||
        (rc = w.mp.ensure(N * 8)))
        return rc;
    ORBFE_HIP(hipMemcpy(w.kps.p, kps, N * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.desc.p, desc, N * 32, hipMemcpyHostToDevice));
    if (mp_index) ORBFE_HIP(hipMemcpy(w.mp.p, mp_index, N * 8, hipMemcpyHostToDevice));
    if ((rc = orbfe_keyframe_features_pack_device(w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(), mp_index ? w.mp.as<uint64_t>() : nullptr,
                                                  nullptr, nullptr, 1, n, w.file.as<uint8_t>(), nullptr)))
        return rc;
    ORBFE_HIP(hipMemcpy(out, w.file.p, N * ORBFE_KF_FEATURE_BYTES, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_keyframe_features_unpack(const uint8_t* in, int n, orbfe_keypoint* kps, uint8_t* desc, uint64_t* mp_index, int device)
{
    if (n < 0 || (n && (!in || !kps || !desc))) return fail(ORBFE_ERR_INVALID, "orbfe_keyframe_features_unpack: invalid argument");
    int rc = use_device(device);
    if (rc || n == 0) return rc;
    KfWorkspace& w = kf_ws();
    const size_t N = (size_t)n;
    if ((rc = w.file.ensure(N * ORBFE_KF_FEATURE_BYTES)) || (rc = w.kps.ensure(N * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure(N * 32)) ||
        (rc = w.mp.ensure(N * 8)) || (rc = w.bad.ensure(16)))
        return rc;
    ORBFE_HIP(hipMemcpy(w.file.p, in, N * ORBFE_KF_FEATURE_BYTES, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemset(w.bad.p, 0, 4));
    if ((rc = orbfe_keyframe_features_unpack_device(w.file.as<uint8_t>(), nullptr, nullptr, 1, n, w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(),
                                                    mp_index ? w.mp.as<uint64_t>() : nullptr, w.bad.as<int32_t>(), nullptr)))
        return rc;
    int32_t bad = 0;
    ORBFE_HIP(hipMemcpy(&bad, w.bad.p, 4, hipMemcpyDeviceToHost));
    if (bad) return fail(ORBFE_ERR_INVALID, "orbfe_keyframe_features_unpack: %d records with a descriptor length other than 32", bad);
    ORBFE_HIP(hipMemcpy(kps, w.kps.p, N * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(desc, w.desc.p, N * 32, hipMemcpyDeviceToHost));
    if (mp_index) ORBFE_HIP(hipMemcpy(mp_index, w.mp.p, N * 8, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

} // extern "C"
