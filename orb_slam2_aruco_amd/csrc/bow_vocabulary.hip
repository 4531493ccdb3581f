// DBoW2 vocabulary transform on gfx950: what Frame::ComputeBoW (reference src/Frame.cc:348-355) asks of
// ORBVocabulary::transform(features, BowVector, FeatureVector, levelsup = 4) -- Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h
// :1127-1194 (vectors) and :1218-1259 (tree descent), FORB.cpp:81-101 (distance), BowVector.cpp:32-89, FeatureVector.cpp:30-45.
//
// Layout in HBM: the tree is stored children-first -- node -> {first child slot, child count}; a slot holds the child's
// node id and its 32-byte descriptor, so the k candidates of one descent step are one contiguous 32k-byte read (ORBvoc:
// k = 10, L = 6, 1.08 M nodes = 35 MB of descriptors, resident in the 256 MB infinity cache after the first frames).
// k_bow_descend: one lane per feature, 8 dwords of query in registers, L dependent steps of <= k popcount distances.
// k_bow_vectors: one workgroup per frame; (word, feature) and (node, feature) keys are bitonic-sorted in LDS and the
// std::map semantics are recovered from the sorted runs: weights of a word are added in feature order, the L1 / L2 norm
// is accumulated in word order by one lane, exactly the order the reference's map iteration produces.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "orbfe_common.hpp"
#include "wave_dpp.hpp"

using namespace orbfe;

#define BV_THREADS 1024
#define BV_MAX 4096 // features per frame the vector kernel sorts in LDS
#define BV_LDS_BYTES (BV_MAX * (8 + 8 + 4))

struct orbfe_vocabulary {
    int device = 0;
    int k = 0, L = 0, scoring = 0, weighting = 0, nnodes = 0, nwords = 0;
    DevBuf info, child_node, child_desc, word_id, weight;                     // the tree
    DevBuf w_desc, w_n, w_word, w_nid, w_weight;                              // host-pointer calls: staging
    DevBuf w_bw, w_bv, w_nb, w_fn, w_fo, w_ff, w_nf;
    std::mutex staging; // ComputeBoW is called from the Tracking, LocalMapping and LoopClosing threads on ONE vocabulary
};

namespace {

struct VocView {
    const int2* info;           // node -> (first child slot, child count)
    const int32_t* child_node;  // slot -> node id
    const uint4* child_desc;    // slot -> 2 x uint4
    const int32_t* word_id;     // node -> word id (0 for inner nodes, as the reference's Node() leaves it)
    const double* weight;       // node -> weight
    int L;
};

// one lane per feature: the descent of TemplatedVocabulary.h:1218-1259
__global__ __launch_bounds__(256) void k_bow_descend(const uint8_t* __restrict__ desc, const int32_t* __restrict__ d_n, int capacity,
                                                     VocView v, int levelsup, int32_t* __restrict__ word, int32_t* __restrict__ nid_out,
                                                     double* __restrict__ weight)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = d_n ? min(d_n[f], capacity) : capacity;
    if (i >= n) return;
    const size_t at = (size_t)f * capacity + i;
    const uint4* q4 = (const uint4*)(desc + at * 32);
    const uint4 qa = q4[0], qb = q4[1];
    const int nid_level = v.L - levelsup;
    int node = 0, level = 0, nid = 0;
    bool nid_set = nid_level <= 0; // :1228 root
    for (;;) {
        const int2 c = v.info[node];
        if (c.y == 0) break; // isLeaf(): no children
        int best = 0x7fffffff, bi = 0;
        for (int j = 0; j < c.y; j++) {
            const uint4 a = v.child_desc[2 * (size_t)(c.x + j)], b = v.child_desc[2 * (size_t)(c.x + j) + 1];
            const int d = __popc(a.x ^ qa.x) + __popc(a.y ^ qa.y) + __popc(a.z ^ qa.z) + __popc(a.w ^ qa.w) + __popc(b.x ^ qb.x) +
                          __popc(b.y ^ qb.y) + __popc(b.z ^ qb.z) + __popc(b.w ^ qb.w);
            if (d < best) { best = d; bi = j; } // strict: the first of equal children stays (:1243)
        }
        node = v.child_node[c.x + bi];
        ++level;
        if (level == nid_level) { nid = node; nid_set = true; }
    }
    // a leaf above level L - levelsup: the reference leaves the caller's NodeId uninitialised; defined here as the leaf
    if (!nid_set) nid = node;
    word[at] = v.word_id[node];
    nid_out[at] = nid;
    weight[at] = v.weight[node];
}

// ascending bitonic sort of npow2 keys in LDS
__device__ inline void bitonic_sort_u64(unsigned long long* key, int npow2)
{
    for (int k = 2; k <= npow2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npow2 >> 1); t += BV_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
                const unsigned long long a = key[i], b = key[p];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { key[i] = b; key[p] = a; }
            }
            __syncthreads();
        }
}

// exclusive prefix count of run heads among the first m sorted keys; returns the number of runs, rank[] gets every head's run index
__device__ inline int rank_run_heads(const unsigned long long* key, int m, int* rank, int* s_wave /*BV_THREADS/64 + 1*/)
{
    const int per = (m + BV_THREADS - 1) / BV_THREADS;
    const int lo = min(m, (int)threadIdx.x * per), hi = min(m, lo + per);
    int c = 0;
    for (int p = lo; p < hi; p++) c += (p == 0 || (key[p] >> 32) != (key[p - 1] >> 32)) ? 1 : 0;
    const int incl = wave_incl_scan_add(c);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 0; w < BV_THREADS / 64; w++) { const int t = s_wave[w]; s_wave[w] = acc; acc += t; }
        s_wave[BV_THREADS / 64] = acc;
    }
    __syncthreads();
    int r = s_wave[wave] + incl - c;
    for (int p = lo; p < hi; p++)
        if (p == 0 || (key[p] >> 32) != (key[p - 1] >> 32)) rank[p] = r++;
        else rank[p] = -1;
    const int total = s_wave[BV_THREADS / 64];
    __syncthreads();
    return total;
}

// one workgroup per frame: BowVector and FeatureVector from the per-feature (word, node, weight)
__global__ __launch_bounds__(BV_THREADS) void k_bow_vectors(const int32_t* __restrict__ word, const int32_t* __restrict__ nid,
                                                            const double* __restrict__ weight, const int32_t* __restrict__ d_n,
                                                            int capacity, int weighting, int norm /*0 none, 1 L1, 2 L2*/,
                                                            uint32_t* __restrict__ bow_word, double* __restrict__ bow_value,
                                                            int32_t* __restrict__ nbow, uint32_t* __restrict__ fv_node,
                                                            int32_t* __restrict__ fv_offset, uint32_t* __restrict__ fv_feature,
                                                            int32_t* __restrict__ nfv)
{
    extern __shared__ unsigned long long s_dyn[]; // BV_LDS_BYTES: keys | values | ranks
    unsigned long long* s_key = s_dyn;
    double* s_val = (double*)(s_dyn + BV_MAX);
    int* s_rank = (int*)(s_dyn + 2 * BV_MAX);
    __shared__ int s_wave[BV_THREADS / 64 + 1];
    __shared__ int s_m;
    __shared__ double s_norm;
    const int f = blockIdx.x;
    const int n = min(d_n ? d_n[f] : capacity, min(capacity, BV_MAX));
    const size_t base = (size_t)f * capacity;
    int npow2 = 2;
    while (npow2 < n) npow2 <<= 1;
    if (threadIdx.x == 0) s_m = 0;
    __syncthreads();
    // ---- BowVector: runs of equal word ids, features ascending inside a run
    int mine = 0;
    for (int i = threadIdx.x; i < npow2; i += BV_THREADS) {
        const bool ok = i < n && weight[base + i] > 0; // "not stopped" (:1156)
        s_key[i] = ok ? ((unsigned long long)(uint32_t)word[base + i] << 32) | (uint32_t)i : ~0ull;
        mine += ok;
    }
    mine = wave_sum(mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_m, mine);
    __syncthreads();
    const int m = s_m;
    bitonic_sort_u64(s_key, npow2);
    const int nb = rank_run_heads(s_key, m, s_rank, s_wave);
    for (int p = threadIdx.x; p < m; p += BV_THREADS) {
        const int r = s_rank[p];
        if (r < 0) continue;
        const uint32_t wd = (uint32_t)(s_key[p] >> 32);
        double v = weight[base + (uint32_t)s_key[p]];
        if (weighting <= 1) // TF_IDF, TF: addWeight in feature order (BowVector.cpp:32-45); IDF, BINARY: addIfNotExist
            for (int t = p + 1; t < m && (uint32_t)(s_key[t] >> 32) == wd; t++) v += weight[base + (uint32_t)s_key[t]];
        s_val[r] = v;
        bow_word[base + r] = wd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double nrm = 0.0;
        if (norm == 1) {
            for (int r = 0; r < nb; r++) nrm += fabs(s_val[r]); // map order = ascending word id (BowVector.cpp:65-69)
        } else if (norm == 2) {
            for (int r = 0; r < nb; r++) nrm += s_val[r] * s_val[r];
            nrm = sqrt(nrm);
        }
        s_norm = nrm;
        nbow[f] = nb;
    }
    __syncthreads();
    {
        const double nrm = s_norm, nd = (double)nb;
        for (int r = threadIdx.x; r < nb; r += BV_THREADS) {
            double v = s_val[r];
            if (norm == 0 && weighting <= 1) v /= nd; // :1163-1169
            if (norm != 0 && nrm > 0.0) v /= nrm;     // BowVector.cpp:78-82
            bow_value[base + r] = v;
        }
    }
    __syncthreads();
    // ---- FeatureVector: runs of equal node ids
    for (int i = threadIdx.x; i < npow2; i += BV_THREADS) {
        const bool ok = i < n && weight[base + i] > 0;
        s_key[i] = ok ? ((unsigned long long)(uint32_t)nid[base + i] << 32) | (uint32_t)i : ~0ull;
    }
    __syncthreads();
    bitonic_sort_u64(s_key, npow2);
    const int nf = rank_run_heads(s_key, m, s_rank, s_wave);
    for (int p = threadIdx.x; p < m; p += BV_THREADS) {
        fv_feature[base + p] = (uint32_t)s_key[p];
        const int r = s_rank[p];
        if (r >= 0) {
            fv_node[base + r] = (uint32_t)(s_key[p] >> 32);
            fv_offset[(size_t)f * (capacity + 1) + r] = p;
        }
    }
    if (threadIdx.x == 0) {
        fv_offset[(size_t)f * (capacity + 1) + nf] = m;
        nfv[f] = nf;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ORBmatcher::SearchByBoW (reference src/ORBmatcher.cc:159-292 KeyFrame-Frame, :526-659 KeyFrame-KeyFrame) over the
// FeatureVectors the transform above produced.  One workgroup per pair of frames.  Features of side 2 belong to exactly one
// node, so the "already matched" coupling of the reference's loop (:213-214, :586) never crosses nodes: a wave owns a common
// node, walks side 1's features of that node in order and spreads side 2's over its lanes; best / second-best follow the
// sequential scan (first minimum wins, the second counts duplicates).  The rotation histogram is LDS counters.
#define SB_THREADS 1024
#define SB_MAX 4096 // features per frame (LDS match tables)

struct BowFrames {
    const orbfe_keypoint* kps;
    const uint8_t* desc;
    const uint8_t* valid; // "has a map point that is not bad"; may be NULL
    const int32_t* n;
    const uint32_t* fv_node;
    const int32_t* fv_off; // blocks of capacity + 1
    const uint32_t* fv_feat;
    const int32_t* nfv;
    int capacity;
};

// SearchForTriangulation (ORBmatcher.cc:661-827, monocular) shares the node pairing and the histogram: tri.F12 != NULL selects
// its candidate rule -- no taken-feature coupling (the reference never sets vbMatched2), candidates pass dist <= TH_LOW, the
// epipole distance gate (:752-757) and CheckDistEpipolarLine (:139-157); the last of the equally near candidates wins (:749).
struct TriParams {
    const float* F12;      // npairs x 9, row major
    const float* epipole;  // npairs x 2 (ex, ey), :669-675
    float scale[16], sigma2[16]; // pKF2->mvScaleFactors, mvLevelSigma2
};

__global__ __launch_bounds__(SB_THREADS) void k_search_by_bow(BowFrames F, const int32_t* __restrict__ pair1, const int32_t* __restrict__ pair2,
                                                             int use_valid2, float nnratio, int check_ori, int accept_max, float factor,
                                                             int32_t* __restrict__ match12, int32_t* __restrict__ match21,
                                                             int32_t* __restrict__ nmatches, TriParams tri)
{
    const bool tri_mode = tri.F12 != nullptr;
    __shared__ short s_m12[SB_MAX], s_m21[SB_MAX];
    __shared__ signed char s_bin[SB_MAX];
    __shared__ int s_hist[30], s_nm, s_ind[3];
    const int p = blockIdx.x, f1 = pair1 ? pair1[p] : p, f2 = pair2 ? pair2[p] : p + 1;
    const int cap = F.capacity;
    const int n1 = min(min(F.n[f1], cap), SB_MAX), n2 = min(min(F.n[f2], cap), SB_MAX);
    for (int i = threadIdx.x; i < SB_MAX; i += SB_THREADS) { s_m12[i] = -1; s_m21[i] = -1; s_bin[i] = -1; }
    if (threadIdx.x < 30) s_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_nm = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nfv1 = F.nfv[f1], nfv2 = F.nfv[f2];
    const uint32_t *node1 = F.fv_node + (size_t)f1 * cap, *node2 = F.fv_node + (size_t)f2 * cap;
    const int32_t *off1 = F.fv_off + (size_t)f1 * (cap + 1), *off2 = F.fv_off + (size_t)f2 * (cap + 1);
    const uint32_t *feat1 = F.fv_feat + (size_t)f1 * cap, *feat2 = F.fv_feat + (size_t)f2 * cap;
    const uint4 *d1v = (const uint4*)(F.desc + (size_t)f1 * cap * 32), *d2v = (const uint4*)(F.desc + (size_t)f2 * cap * 32);
    const uint8_t *valid1 = F.valid ? F.valid + (size_t)f1 * cap : nullptr, *valid2 = (F.valid && use_valid2) ? F.valid + (size_t)f2 * cap : nullptr;
    const orbfe_keypoint *k1 = F.kps + (size_t)f1 * cap, *k2 = F.kps + (size_t)f2 * cap;
    int nm = 0;
    for (int a = wave; a < nfv1; a += SB_THREADS / 64) {
        const uint32_t node = node1[a];
        int lo = 0, hi = nfv2; // lower_bound in side 2's node list (:279-286)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (node2[mid] < node) lo = mid + 1; else hi = mid;
        }
        if (lo >= nfv2 || node2[lo] != node) continue;
        const int o1 = off1[a], e1 = off1[a + 1], o2 = off2[lo], e2 = off2[lo + 1];
        for (int p1 = o1; p1 < e1; p1++) {
            const int idx1 = (int)feat1[p1];
            if (idx1 >= n1 || (valid1 && !valid1[idx1])) continue;
            const uint4 qa = d1v[2 * idx1], qb = d1v[2 * idx1 + 1];
            int bestd = 256, second = 256, key = (256 << 16) | 0xffff;
            float la = 0.0f, lb = 0.0f, lc = 0.0f, lden = 0.0f, ex = 0.0f, ey = 0.0f;
            if (tri_mode) { // epipolar line of kp1 in the second image, l = x1' F12 (:142-145)
                const float* Fm = tri.F12 + 9 * (size_t)p;
                const float x1 = k1[idx1].x, y1 = k1[idx1].y;
                la = __fadd_rn(__fadd_rn(__fmul_rn(x1, Fm[0]), __fmul_rn(y1, Fm[3])), Fm[6]);
                lb = __fadd_rn(__fadd_rn(__fmul_rn(x1, Fm[1]), __fmul_rn(y1, Fm[4])), Fm[7]);
                lc = __fadd_rn(__fadd_rn(__fmul_rn(x1, Fm[2]), __fmul_rn(y1, Fm[5])), Fm[8]);
                lden = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
                ex = tri.epipole[2 * p]; ey = tri.epipole[2 * p + 1];
            }
            for (int p2 = o2 + lane; p2 < e2; p2 += 64) {
                const int idx2 = (int)feat2[p2];
                if (idx2 >= n2 || (!tri_mode && s_m21[idx2] >= 0) || (valid2 && !valid2[idx2])) continue;
                const uint4 ta = d2v[2 * idx2], tb = d2v[2 * idx2 + 1];
                const int dist = __popc(ta.x ^ qa.x) + __popc(ta.y ^ qa.y) + __popc(ta.z ^ qa.z) + __popc(ta.w ^ qa.w) +
                                 __popc(tb.x ^ qb.x) + __popc(tb.y ^ qb.y) + __popc(tb.z ^ qb.z) + __popc(tb.w ^ qb.w);
                if (tri_mode) {
                    if (dist > accept_max) continue; // :749
                    const orbfe_keypoint kp2 = k2[idx2];
                    const int oct = min(max(kp2.octave, 0), 15);
                    const float dex = ex - kp2.x, dey = ey - kp2.y;
                    if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.0f, tri.scale[oct])) continue; // :752-757
                    const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, kp2.x), __fmul_rn(lb, kp2.y)), lc);
                    if (lden == 0.0f) continue;
                    const float dsqr = __fdiv_rn(__fmul_rn(num, num), lden);
                    if (!((double)dsqr < 3.84 * (double)tri.sigma2[oct])) continue; // :156
                    const int k = (dist << 16) | (0xffff - (p2 - o2)); // "dist > bestDist" skips: of equal distances the last stays
                    if (k < key) { key = k; bestd = dist; }
                    continue;
                }
                if (dist < bestd) { second = bestd; bestd = dist; key = (dist << 16) | (p2 - o2); }
                else if (dist < second) second = dist;
            }
            const int B = wave_min(key);
            const int S = wave_min(key == B ? second : bestd);
            const int bestDist1 = B >> 16;
            if (tri_mode ? (B >> 16) <= accept_max : (bestDist1 <= accept_max && (float)bestDist1 < __fmul_rn(nnratio, (float)S))) {
                const int idx2 = (int)feat2[o2 + (tri_mode ? 0xffff - (B & 0xffff) : (B & 0xffff))];
                if (lane == 0) {
                    s_m12[idx1] = (short)idx2;
                    if (!tri_mode) s_m21[idx2] = (short)idx1;
                    if (check_ori) {
                        float rot = k1[idx1].angle - k2[idx2].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(__fmul_rn(rot, factor));
                        if (bin == 30) bin = 0;
                        bin = min(max(bin, 0), 29); // the reference asserts this range
                        s_bin[idx1] = (signed char)bin;
                        atomicAdd(&s_hist[bin], 1);
                    }
                }
                nm++;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane == 0 && nm) atomicAdd(&s_nm, nm);
    __syncthreads();
    if (check_ori) {
        if (threadIdx.x == 0) {
            int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < 30; i++) { // ComputeThreeMaxima, ORBmatcher.cc:1605-1646
                const int s = s_hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
            s_ind[0] = ind1; s_ind[1] = ind2; s_ind[2] = ind3;
        }
        __syncthreads();
        int removed = 0;
        for (int i = threadIdx.x; i < n1; i += SB_THREADS) {
            const int bin = s_bin[i];
            if (bin >= 0 && bin != s_ind[0] && bin != s_ind[1] && bin != s_ind[2]) {
                if (!tri_mode) s_m21[s_m12[i]] = -1;
                s_m12[i] = -1;
                removed++;
            }
        }
        removed = wave_sum(removed);
        if (lane == 0 && removed) atomicSub(&s_nm, removed);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n1; i += SB_THREADS) match12[(size_t)p * cap + i] = s_m12[i];
    for (int i = threadIdx.x; i < n2; i += SB_THREADS) match21[(size_t)p * cap + i] = s_m21[i];
    if (threadIdx.x == 0) nmatches[p] = s_nm;
}

struct BowMatchWorkspace {
    DevBuf kps, desc, valid, n, fn, fo, ff, nf, m12, m21, nm;
};
thread_local ThreadWorkspaces<BowMatchWorkspace> tl_bow_ws; // per (thread, device): host-pointer entry points only

int norm_of(int scoring) { return scoring == 5 ? 0 : scoring == 1 ? 2 : 1; } // ScoringObject.h:74-91

// builds the device tree from nodes given in file order (node 0 = root, implicit)
int upload(orbfe_vocabulary* v, const std::vector<int32_t>& parent, const std::vector<uint8_t>& is_leaf,
           const std::vector<uint8_t>& desc, const std::vector<double>& weight)
{
    const int nn = (int)parent.size(); // incl. root at index 0
    std::vector<int32_t> count(nn, 0), begin(nn, 0), wid(nn, 0);
    for (int i = 1; i < nn; i++) {
        if (parent[i] < 0 || parent[i] >= i) return fail(ORBFE_ERR_INVALID, "vocabulary: node %d has parent %d (must be an earlier node)", i, parent[i]);
        count[parent[i]]++;
    }
    int total = 0;
    for (int i = 0; i < nn; i++) { begin[i] = total; total += count[i]; }
    std::vector<int32_t> fill(begin), cnode(std::max(total, 1));
    std::vector<uint8_t> cdesc((size_t)std::max(total, 1) * 32);
    int nwords = 0;
    for (int i = 1; i < nn; i++) {
        const int slot = fill[parent[i]]++; // children keep their file order (push_back, :1389)
        cnode[slot] = i;
        memcpy(&cdesc[(size_t)slot * 32], &desc[(size_t)i * 32], 32);
        if (is_leaf[i]) wid[i] = nwords++;
    }
    std::vector<int2> info(nn);
    for (int i = 0; i < nn; i++) info[i] = make_int2(begin[i], count[i]);
    v->nnodes = nn;
    v->nwords = nwords;
    int rc;
    if ((rc = v->info.ensure((size_t)nn * 8)) || (rc = v->child_node.ensure(cnode.size() * 4)) || (rc = v->child_desc.ensure(cdesc.size())) ||
        (rc = v->word_id.ensure((size_t)nn * 4)) || (rc = v->weight.ensure((size_t)nn * 8)))
        return rc;
    ORBFE_HIP(hipMemcpy(v->info.p, info.data(), (size_t)nn * 8, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(v->child_node.p, cnode.data(), cnode.size() * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(v->child_desc.p, cdesc.data(), cdesc.size(), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(v->word_id.p, wid.data(), (size_t)nn * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(v->weight.p, weight.data(), (size_t)nn * 8, hipMemcpyHostToDevice));
    return ORBFE_OK;
}

int header_ok(int k, int L, int scoring, int weighting)
{
    // the checks of loadFromTextFile (:1357-1361)
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)
        return fail(ORBFE_ERR_INVALID, "Vocabulary loading failure: This is not a correct text file! (k %d, L %d, scoring %d, weighting %d)", k, L,
                    scoring, weighting);
    return ORBFE_OK;
}

VocView view(const orbfe_vocabulary* v)
{
    return VocView{v->info.as<int2>(), v->child_node.as<int32_t>(), v->child_desc.as<uint4>(), v->word_id.as<int32_t>(),
                   v->weight.as<double>(), v->L};
}

void release(orbfe_vocabulary* v)
{
    for (DevBuf* b : {&v->info, &v->child_node, &v->child_desc, &v->word_id, &v->weight, &v->w_desc, &v->w_n, &v->w_word, &v->w_nid,
                      &v->w_weight, &v->w_bw, &v->w_bv, &v->w_nb, &v->w_fn, &v->w_fo, &v->w_ff, &v->w_nf})
        b->release();
}

} // namespace

extern "C" {

orbfe_vocabulary* orbfe_vocabulary_create(int k, int L, int scoring, int weighting, int nnodes, const int32_t* parent,
                                          const uint8_t* is_leaf, const uint8_t* descriptors, const double* weights, int device)
{
    if (nnodes < 0 || (nnodes && (!parent || !is_leaf || !descriptors || !weights))) {
        fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_create: invalid argument");
        return nullptr;
    }
    if (header_ok(k, L, scoring, weighting) || use_device(device)) return nullptr;
    orbfe_vocabulary* v = new orbfe_vocabulary();
    v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    std::vector<int32_t> par(nnodes + 1, 0);
    std::vector<uint8_t> leaf(nnodes + 1, 0), desc((size_t)(nnodes + 1) * 32, 0);
    std::vector<double> w(nnodes + 1, 0.0);
    for (int i = 0; i < nnodes; i++) {
        par[i + 1] = parent[i]; leaf[i + 1] = is_leaf[i]; w[i + 1] = weights[i];
        memcpy(&desc[(size_t)(i + 1) * 32], descriptors + (size_t)i * 32, 32);
    }
    if (upload(v, par, leaf, desc, w)) { release(v); delete v; return nullptr; }
    return v;
}

// TemplatedVocabulary::loadFromTextFile (:1338-1425): "k L scoring weighting", then one node per line:
// "parent isLeaf d0 .. d31 weight".  An empty line -- in particular the one `while(!f.eof())` reads after the final
// newline of every file saveToTextFile wrote -- becomes a node as it does there: child of the root, not a word, weight 0,
// descriptor unset (zeros here).
orbfe_vocabulary* orbfe_vocabulary_load_text(const char* filename, int device)
{
    if (!filename) { fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_load_text: null filename"); return nullptr; }
    FILE* fp = fopen(filename, "rb");
    if (!fp) { fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_load_text: cannot open %s", filename); return nullptr; }
    std::string buf;
    {
        char tmp[1 << 16];
        size_t got;
        while ((got = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.append(tmp, got);
        fclose(fp);
    }
    const char* p = buf.c_str();
    const char* end = p + buf.size();
    auto line_end = [&](const char* s) { const char* e = (const char*)memchr(s, '\n', end - s); return e ? e : end; };
    const char* le = line_end(p);
    int k = -1, L = -1, n1 = -1, n2 = -1;
    {
        std::string hdr(p, le);
        if (sscanf(hdr.c_str(), "%d %d %d %d", &k, &L, &n1, &n2) != 4) { fail(ORBFE_ERR_INVALID, "Vocabulary loading failure: This is not a correct text file!"); return nullptr; }
    }
    if (header_ok(k, L, n1, n2) || use_device(device)) return nullptr;
    std::vector<int32_t> par(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> w(1, 0.0);
    bool more = le < end; // a '\n' ended the header: getline will be called again
    p = le < end ? le + 1 : end;
    int lineno = 1;
    while (more) {
        le = line_end(p);
        more = le < end;
        lineno++;
        int32_t pid = 0, isleaf = 0;
        uint8_t d[32] = {0};
        double weight = 0.0;
        const char* s = p;
        while (s < le && (*s == ' ' || *s == '\t' || *s == '\r')) s++;
        if (s < le) {
            char* e;
            pid = (int32_t)strtol(s, &e, 10); bool ok = e != s; s = e;
            isleaf = (int32_t)strtol(s, &e, 10); ok = ok && e != s; s = e;
            for (int i = 0; i < 32 && ok; i++) { long b = strtol(s, &e, 10); ok = e != s && e <= le; s = e; d[i] = (uint8_t)b; }
            if (ok) { weight = strtod(s, &e); ok = e != s && e <= le; }
            if (!ok || pid < 0 || pid >= (int)par.size()) {
                fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_load_text: %s:%d is not a node line", filename, lineno);
                return nullptr;
            }
        }
        par.push_back(pid); leaf.push_back(isleaf > 0); w.push_back(weight);
        desc.insert(desc.end(), d, d + 32);
        p = more ? le + 1 : end;
    }
    orbfe_vocabulary* v = new orbfe_vocabulary();
    v->device = device; v->k = k; v->L = L; v->scoring = n1; v->weighting = n2;
    if (upload(v, par, leaf, desc, w)) { release(v); delete v; return nullptr; }
    return v;
}

void orbfe_vocabulary_destroy(orbfe_vocabulary* v)
{
    if (!v) return;
    (void)use_device(v->device);
    release(v);
    delete v;
}

int orbfe_vocabulary_info(const orbfe_vocabulary* v, int32_t* out6)
{
    if (!v || !out6) return fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_info: null argument");
    out6[0] = v->k; out6[1] = v->L; out6[2] = v->scoring; out6[3] = v->weighting; out6[4] = v->nnodes; out6[5] = v->nwords;
    return ORBFE_OK;
}

int orbfe_vocabulary_transform_batch_device(orbfe_vocabulary* v, const uint8_t* d_desc, const int32_t* d_n, int capacity, int nframes,
                                            int levelsup, int32_t* d_word, int32_t* d_node, double* d_weight, uint32_t* d_bow_word,
                                            double* d_bow_value, int32_t* d_nbow, uint32_t* d_fv_node, int32_t* d_fv_offset,
                                            uint32_t* d_fv_feature, int32_t* d_nfv, void* stream)
{
    if (!v || nframes < 0 || capacity < 0 || (nframes && capacity && (!d_desc || !d_word || !d_node || !d_weight)))
        return fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_transform_batch_device: invalid argument");
    const bool vectors = d_bow_word || d_bow_value || d_nbow || d_fv_node || d_fv_offset || d_fv_feature || d_nfv;
    if (vectors && !(d_bow_word && d_bow_value && d_nbow && d_fv_node && d_fv_offset && d_fv_feature && d_nfv))
        return fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_transform_batch_device: the vector outputs are all or none");
    if (vectors && capacity > BV_MAX) return fail(ORBFE_ERR_CAPACITY, "orbfe_vocabulary_transform: at most %d features per frame", BV_MAX);
    int rc = use_device(v->device);
    if (rc) return rc;
    if (nframes == 0) return ORBFE_OK;
    hipStream_t s = (hipStream_t)stream;
    if (v->nwords == 0 || capacity == 0) { // empty(): both vectors stay empty (:1134-1137)
        if (vectors) {
            ORBFE_HIP(hipMemsetAsync(d_nbow, 0, (size_t)nframes * 4, s));
            ORBFE_HIP(hipMemsetAsync(d_nfv, 0, (size_t)nframes * 4, s));
            ORBFE_HIP(hipMemsetAsync(d_fv_offset, 0, (size_t)nframes * (capacity + 1) * 4, s));
        }
        if (capacity) {
            ORBFE_HIP(hipMemsetAsync(d_word, 0, (size_t)nframes * capacity * 4, s));
            ORBFE_HIP(hipMemsetAsync(d_node, 0, (size_t)nframes * capacity * 4, s));
            ORBFE_HIP(hipMemsetAsync(d_weight, 0, (size_t)nframes * capacity * 8, s));
        }
        return ORBFE_OK;
    }
    hipLaunchKernelGGL(k_bow_descend, dim3((capacity + 255) / 256, nframes), dim3(256), 0, s, d_desc, d_n, capacity, view(v), levelsup,
                       d_word, d_node, d_weight);
    ORBFE_HIP(hipGetLastError());
    if (vectors) {
        { int rc_lds_ = ensure_dyn_lds(reinterpret_cast<const void*>(&k_bow_vectors), (size_t)BV_LDS_BYTES); if (rc_lds_) return rc_lds_; }
        hipLaunchKernelGGL(k_bow_vectors, dim3(nframes), dim3(BV_THREADS), BV_LDS_BYTES, s, d_word, d_node, d_weight, d_n, capacity, v->weighting,
                           norm_of(v->scoring), d_bow_word, d_bow_value, d_nbow, d_fv_node, d_fv_offset, d_fv_feature, d_nfv);
        ORBFE_HIP(hipGetLastError());
    }
    return ORBFE_OK;
}

int orbfe_vocabulary_transform(orbfe_vocabulary* v, const uint8_t* desc, int n, int levelsup, int32_t* word_id, int32_t* node_id,
                               double* weight, uint32_t* bow_word, double* bow_value, int32_t* nbow, uint32_t* fv_node,
                               int32_t* fv_offset, uint32_t* fv_feature, int32_t* nfv)
{
    if (!v || n < 0 || (n && !desc)) return fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_transform: invalid argument");
    const bool vectors = bow_word || bow_value || nbow || fv_node || fv_offset || fv_feature || nfv;
    if (vectors && !(bow_word && bow_value && nbow && fv_node && fv_offset && fv_feature && nfv))
        return fail(ORBFE_ERR_INVALID, "orbfe_vocabulary_transform: the vector outputs are all or none");
    int rc = use_device(v->device);
    if (rc) return rc;
    if (vectors) { *nbow = 0; *nfv = 0; fv_offset[0] = 0; }
    if (n == 0) return ORBFE_OK;
    std::lock_guard<std::mutex> lock(v->staging); // the handle's staging buffers are shared by the calling threads
    const size_t N = (size_t)n;
    if ((rc = v->w_desc.ensure(N * 32)) || (rc = v->w_word.ensure(N * 4)) || (rc = v->w_nid.ensure(N * 4)) || (rc = v->w_weight.ensure(N * 8)) ||
        (rc = v->w_bw.ensure(N * 4)) || (rc = v->w_bv.ensure(N * 8)) || (rc = v->w_nb.ensure(16)) || (rc = v->w_fn.ensure(N * 4)) ||
        (rc = v->w_fo.ensure((N + 1) * 4)) || (rc = v->w_ff.ensure(N * 4)) || (rc = v->w_nf.ensure(16)))
        return rc;
    ORBFE_HIP(hipMemcpy(v->w_desc.p, desc, N * 32, hipMemcpyHostToDevice));
    rc = orbfe_vocabulary_transform_batch_device(
        v, v->w_desc.as<uint8_t>(), nullptr, n, 1, levelsup, v->w_word.as<int32_t>(), v->w_nid.as<int32_t>(), v->w_weight.as<double>(),
        vectors ? v->w_bw.as<uint32_t>() : nullptr, vectors ? v->w_bv.as<double>() : nullptr, vectors ? v->w_nb.as<int32_t>() : nullptr,
        vectors ? v->w_fn.as<uint32_t>() : nullptr, vectors ? v->w_fo.as<int32_t>() : nullptr, vectors ? v->w_ff.as<uint32_t>() : nullptr,
        vectors ? v->w_nf.as<int32_t>() : nullptr, nullptr);
    if (rc) return rc;
    if (word_id) ORBFE_HIP(hipMemcpy(word_id, v->w_word.p, N * 4, hipMemcpyDeviceToHost));
    if (node_id) ORBFE_HIP(hipMemcpy(node_id, v->w_nid.p, N * 4, hipMemcpyDeviceToHost));
    if (weight) ORBFE_HIP(hipMemcpy(weight, v->w_weight.p, N * 8, hipMemcpyDeviceToHost));
    if (vectors) {
        ORBFE_HIP(hipMemcpy(nbow, v->w_nb.p, 4, hipMemcpyDeviceToHost));
        ORBFE_HIP(hipMemcpy(nfv, v->w_nf.p, 4, hipMemcpyDeviceToHost));
        if (*nbow) {
            ORBFE_HIP(hipMemcpy(bow_word, v->w_bw.p, (size_t)*nbow * 4, hipMemcpyDeviceToHost));
            ORBFE_HIP(hipMemcpy(bow_value, v->w_bv.p, (size_t)*nbow * 8, hipMemcpyDeviceToHost));
        }
        ORBFE_HIP(hipMemcpy(fv_offset, v->w_fo.p, (size_t)(*nfv + 1) * 4, hipMemcpyDeviceToHost));
        if (*nfv) {
            ORBFE_HIP(hipMemcpy(fv_node, v->w_fn.p, (size_t)*nfv * 4, hipMemcpyDeviceToHost));
            ORBFE_HIP(hipMemcpy(fv_feature, v->w_ff.p, (size_t)fv_offset[*nfv] * 4, hipMemcpyDeviceToHost));
        }
    }
    return ORBFE_OK;
}

int orbfe_search_by_bow_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_valid, const int32_t* d_n,
                                     const uint32_t* d_fv_node, const int32_t* d_fv_offset, const uint32_t* d_fv_feature,
                                     const int32_t* d_nfv, int capacity, const int32_t* d_pair1, const int32_t* d_pair2, int npairs,
                                     int use_valid2, float nnratio, int check_orientation, int accept_max, float factor,
                                     int32_t* d_match12, int32_t* d_match21, int32_t* d_nmatches, void* stream)
{
    if (npairs < 0 || capacity <= 0 || !d_kps || !d_desc || !d_n || !d_fv_node || !d_fv_offset || !d_fv_feature || !d_nfv || !d_match12 ||
        !d_match21 || !d_nmatches)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_bow_batch_device: invalid argument");
    if (capacity > SB_MAX) return fail(ORBFE_ERR_CAPACITY, "orbfe_search_by_bow: at most %d features per frame", SB_MAX);
    if (npairs == 0) return ORBFE_OK;
    BowFrames F{d_kps, d_desc, d_valid, d_n, d_fv_node, d_fv_offset, d_fv_feature, d_nfv, capacity};
    hipLaunchKernelGGL(k_search_by_bow, dim3(npairs), dim3(SB_THREADS), 0, (hipStream_t)stream, F, d_pair1, d_pair2, use_valid2, nnratio,
                       check_orientation, accept_max, factor, d_match12, d_match21, d_nmatches, TriParams{});
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_search_for_triangulation_batch_device(const orbfe_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_free, const int32_t* d_n,
                                                const uint32_t* d_fv_node, const int32_t* d_fv_offset, const uint32_t* d_fv_feature,
                                                const int32_t* d_nfv, int capacity, const int32_t* d_pair1, const int32_t* d_pair2, int npairs,
                                                const float* d_F12, const float* d_epipole, const float* scale_factors,
                                                const float* level_sigma2, int nlevels, int check_orientation, int32_t* d_match12,
                                                int32_t* d_scratch21, int32_t* d_nmatches, void* stream)
{
    if (npairs < 0 || capacity <= 0 || !d_kps || !d_desc || !d_n || !d_fv_node || !d_fv_offset || !d_fv_feature || !d_nfv || !d_match12 ||
        !d_scratch21 || !d_nmatches || !d_F12 || !d_epipole || !scale_factors || !level_sigma2 || nlevels < 1 || nlevels > 16)
        return fail(ORBFE_ERR_INVALID, "orbfe_search_for_triangulation_batch_device: invalid argument");
    if (capacity > SB_MAX) return fail(ORBFE_ERR_CAPACITY, "orbfe_search_for_triangulation: at most %d features per frame", SB_MAX);
    if (npairs == 0) return ORBFE_OK;
    TriParams tri{};
    tri.F12 = d_F12; tri.epipole = d_epipole;
    for (int l = 0; l < 16; l++) { tri.scale[l] = scale_factors[std::min(l, nlevels - 1)]; tri.sigma2[l] = level_sigma2[std::min(l, nlevels - 1)]; }
    BowFrames F{d_kps, d_desc, d_free, d_n, d_fv_node, d_fv_offset, d_fv_feature, d_nfv, capacity};
    hipLaunchKernelGGL(k_search_by_bow, dim3(npairs), dim3(SB_THREADS), 0, (hipStream_t)stream, F, d_pair1, d_pair2, 1, 0.0f,
                       check_orientation, 50 /* TH_LOW */, 1.0f / 30 /* :693 */, d_match12, d_scratch21, d_nmatches, tri);
    ORBFE_HIP(hipGetLastError());
    return ORBFE_OK;
}

int orbfe_search_for_triangulation(const orbfe_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_mp1, int n1, const uint32_t* fv_node1,
                                   const int32_t* fv_offset1, const uint32_t* fv_feature1, int nfv1, const orbfe_keypoint* kps2,
                                   const uint8_t* desc2, const uint8_t* has_mp2, int n2, const uint32_t* fv_node2, const int32_t* fv_offset2,
                                   const uint32_t* fv_feature2, int nfv2, const float* F12, float ex, float ey, const float* scale_factors2,
                                   const float* level_sigma2_2, int nlevels, int check_orientation, int32_t* match12, int32_t* nmatches,
                                   int device)
{
    if (n1 < 0 || n2 < 0 || nfv1 < 0 || nfv2 < 0 || nfv1 > n1 || nfv2 > n2 || !nmatches || !F12 || !scale_factors2 || !level_sigma2_2 ||
        nlevels < 1 || nlevels > 16 || (n1 && (!kps1 || !desc1 || !match12)) || (n2 && (!kps2 || !desc2)) ||
        (nfv1 && (!fv_node1 || !fv_offset1 || !fv_feature1)) || (nfv2 && (!fv_node2 || !fv_offset2 || !fv_feature2)))
        return fail(ORBFE_ERR_INVALID, "orbfe_search_for_triangulation: invalid argument");
    if (nfv1 && (fv_offset1[0] != 0 || fv_offset1[nfv1] > n1)) return fail(ORBFE_ERR_INVALID, "orbfe_search_for_triangulation: bad FeatureVector 1");
    if (nfv2 && (fv_offset2[0] != 0 || fv_offset2[nfv2] > n2)) return fail(ORBFE_ERR_INVALID, "orbfe_search_for_triangulation: bad FeatureVector 2");
    int rc = use_device(device);
    if (rc) return rc;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0 || nfv1 == 0 || nfv2 == 0) return ORBFE_OK;
    const int cap = std::max(n1, n2);
    if (cap > SB_MAX) return fail(ORBFE_ERR_CAPACITY, "orbfe_search_for_triangulation: at most %d features per frame", SB_MAX);
    BowMatchWorkspace& w = tl_bow_ws.get();
    const size_t C = (size_t)cap;
    if ((rc = w.kps.ensure(2 * C * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure(2 * C * 32)) || (rc = w.valid.ensure(2 * C)) ||
        (rc = w.n.ensure(16)) || (rc = w.fn.ensure(2 * C * 4)) || (rc = w.fo.ensure(2 * (C + 1) * 4)) || (rc = w.ff.ensure(2 * C * 4)) ||
        (rc = w.nf.ensure(64)) || (rc = w.m12.ensure(C * 4)) || (rc = w.m21.ensure(C * 4)) || (rc = w.nm.ensure(64)))
        return rc;
    std::vector<uint8_t> freef(2 * C, 1); // "no map point yet" (:710-714, :731-735)
    if (has_mp1) for (int i = 0; i < n1; i++) freef[i] = !has_mp1[i];
    if (has_mp2) for (int i = 0; i < n2; i++) freef[C + i] = !has_mp2[i];
    const int32_t nn[2] = {n1, n2}, nf[2] = {nfv1, nfv2};
    const float fe[11] = {F12[0], F12[1], F12[2], F12[3], F12[4], F12[5], F12[6], F12[7], F12[8], ex, ey};
    ORBFE_HIP(hipMemcpy(w.kps.p, kps1, (size_t)n1 * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.kps.as<orbfe_keypoint>() + C, kps2, (size_t)n2 * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.desc.p, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.desc.as<uint8_t>() + C * 32, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.valid.p, freef.data(), 2 * C, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.n.p, nn, 8, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.nf.p, nf, 8, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.nm.as<float>() + 4, fe, sizeof fe, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fn.p, fv_node1, (size_t)nfv1 * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fn.as<uint32_t>() + C, fv_node2, (size_t)nfv2 * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fo.p, fv_offset1, (size_t)(nfv1 + 1) * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fo.as<int32_t>() + C + 1, fv_offset2, (size_t)(nfv2 + 1) * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.ff.p, fv_feature1, (size_t)fv_offset1[nfv1] * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.ff.as<uint32_t>() + C, fv_feature2, (size_t)fv_offset2[nfv2] * 4, hipMemcpyHostToDevice));
    rc = orbfe_search_for_triangulation_batch_device(w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(), w.valid.as<uint8_t>(), w.n.as<int32_t>(),
                                                     w.fn.as<uint32_t>(), w.fo.as<int32_t>(), w.ff.as<uint32_t>(), w.nf.as<int32_t>(), cap, nullptr,
                                                     nullptr, 1, w.nm.as<float>() + 4, w.nm.as<float>() + 13, scale_factors2, level_sigma2_2,
                                                     nlevels, check_orientation, w.m12.as<int32_t>(), w.m21.as<int32_t>(), w.nm.as<int32_t>(),
                                                     nullptr);
    if (rc) return rc;
    ORBFE_HIP(hipMemcpy(match12, w.m12.p, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(nmatches, w.nm.p, 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_search_by_bow(const orbfe_keypoint* kps1, const uint8_t* desc1, const uint8_t* valid1, int n1, const uint32_t* fv_node1,
                        const int32_t* fv_offset1, const uint32_t* fv_feature1, int nfv1, const orbfe_keypoint* kps2,
                        const uint8_t* desc2, const uint8_t* valid2, int n2, const uint32_t* fv_node2, const int32_t* fv_offset2,
                        const uint32_t* fv_feature2, int nfv2, float nnratio, int check_orientation, int accept_max, float factor,
                        int32_t* match12, int32_t* match21, int32_t* nmatches, int device)
{
    if (n1 < 0 || n2 < 0 || nfv1 < 0 || nfv2 < 0 || nfv1 > n1 || nfv2 > n2 || !nmatches || (n1 && (!kps1 || !desc1 || !match12)) ||
        (n2 && (!kps2 || !desc2 || !match21)) || (nfv1 && (!fv_node1 || !fv_offset1 || !fv_feature1)) ||
        (nfv2 && (!fv_node2 || !fv_offset2 || !fv_feature2)))
        return fail(ORBFE_ERR_INVALID, "orbfe_search_by_bow: invalid argument");
    if (nfv1 && (fv_offset1[0] != 0 || fv_offset1[nfv1] > n1)) return fail(ORBFE_ERR_INVALID, "orbfe_search_by_bow: bad FeatureVector 1");
    if (nfv2 && (fv_offset2[0] != 0 || fv_offset2[nfv2] > n2)) return fail(ORBFE_ERR_INVALID, "orbfe_search_by_bow: bad FeatureVector 2");
    int rc = use_device(device);
    if (rc) return rc;
    *nmatches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    for (int i = 0; i < n2; i++) match21[i] = -1;
    if (n1 == 0 || n2 == 0 || nfv1 == 0 || nfv2 == 0) return ORBFE_OK;
    const int cap = std::max(n1, n2);
    if (cap > SB_MAX) return fail(ORBFE_ERR_CAPACITY, "orbfe_search_by_bow: at most %d features per frame", SB_MAX);
    BowMatchWorkspace& w = tl_bow_ws.get();
    const size_t C = (size_t)cap;
    if ((rc = w.kps.ensure(2 * C * sizeof(orbfe_keypoint))) || (rc = w.desc.ensure(2 * C * 32)) || (rc = w.valid.ensure(2 * C)) ||
        (rc = w.n.ensure(16)) || (rc = w.fn.ensure(2 * C * 4)) || (rc = w.fo.ensure(2 * (C + 1) * 4)) || (rc = w.ff.ensure(2 * C * 4)) ||
        (rc = w.nf.ensure(16)) || (rc = w.m12.ensure(C * 4)) || (rc = w.m21.ensure(C * 4)) || (rc = w.nm.ensure(16)))
        return rc;
    const bool any_valid = valid1 || valid2;
    std::vector<uint8_t> valid(2 * C, 1);
    if (valid1) memcpy(valid.data(), valid1, n1);
    if (valid2) memcpy(valid.data() + C, valid2, n2);
    const int32_t nn[2] = {n1, n2}, nf[2] = {nfv1, nfv2};
    ORBFE_HIP(hipMemcpy(w.kps.p, kps1, (size_t)n1 * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.kps.as<orbfe_keypoint>() + C, kps2, (size_t)n2 * sizeof(orbfe_keypoint), hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.desc.p, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.desc.as<uint8_t>() + C * 32, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.valid.p, valid.data(), 2 * C, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.n.p, nn, 8, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.nf.p, nf, 8, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fn.p, fv_node1, (size_t)nfv1 * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fn.as<uint32_t>() + C, fv_node2, (size_t)nfv2 * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fo.p, fv_offset1, (size_t)(nfv1 + 1) * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.fo.as<int32_t>() + C + 1, fv_offset2, (size_t)(nfv2 + 1) * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.ff.p, fv_feature1, (size_t)fv_offset1[nfv1] * 4, hipMemcpyHostToDevice));
    ORBFE_HIP(hipMemcpy(w.ff.as<uint32_t>() + C, fv_feature2, (size_t)fv_offset2[nfv2] * 4, hipMemcpyHostToDevice));
    rc = orbfe_search_by_bow_batch_device(w.kps.as<orbfe_keypoint>(), w.desc.as<uint8_t>(), any_valid ? w.valid.as<uint8_t>() : nullptr,
                                          w.n.as<int32_t>(), w.fn.as<uint32_t>(), w.fo.as<int32_t>(), w.ff.as<uint32_t>(), w.nf.as<int32_t>(),
                                          cap, nullptr, nullptr, 1, valid2 != nullptr, nnratio, check_orientation, accept_max, factor,
                                          w.m12.as<int32_t>(), w.m21.as<int32_t>(), w.nm.as<int32_t>(), nullptr);
    if (rc) return rc;
    ORBFE_HIP(hipMemcpy(match12, w.m12.p, (size_t)n1 * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(match21, w.m21.p, (size_t)n2 * 4, hipMemcpyDeviceToHost));
    ORBFE_HIP(hipMemcpy(nmatches, w.nm.p, 4, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

} // extern "C"
