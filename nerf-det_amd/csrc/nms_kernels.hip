// A15: greedy class-aware axis-aligned 3D NMS with the reference's pick order
// (mmdet3d/core/post_processing/box3d_nms.py:91-138), n <= 4096 candidates (the head feeds <= 3000).
//
//   1. k_nms_sort      one workgroup: bitonic sort of (score, index) in LDS -> `order`, highest score first
//                      (the reference argsorts ascending and pops from the back; ties resolved by higher
//                      index first here -- torch.argsort is unstable, so the reference leaves ties undefined).
//   2. k_nms_mask      n x ceil(n/64) suppression bit matrix over the sorted list: bit (i, j) set when
//                      picking i removes j, i.e. NOT (iou_ij * [class_i == class_j] <= thresh).  NaN IoU
//                      (0/0 on zero-volume pairs) fails the <= and removes, exactly as the reference.
//   3. k_nms_sweep     one wavefront walks the sorted list 64 candidates at a time; the removed-set lives in one 64-bit
//                      word per lane, the walk inside a block is a scalar bit loop, only the winners' rows are fetched.
#include "ndet_common.hpp"

#define NMS_MAX 4096

// n_dev != null: the candidate count lives on the device (no host round trip between the compaction and the NMS); more than n_cap
// candidates make every kernel of the chain do nothing and the packed header report it (the host then takes the synchronous path).
__device__ __forceinline__ int nms_count(int n, const int* __restrict__ n_dev, int n_cap) {
    if (n_dev == nullptr) return n;
    const int m = *n_dev;
    return m > n_cap ? 0 : m;
}

// Bitonic sort of 4096 (score, index) pairs by one workgroup, four consecutive positions per thread: the exchange partner of
// position i in a pass of stride j is i ^ j -- inside the thread for j < 4, in another lane of the same wavefront for j < 256
// (a cross-lane shuffle, no barrier), in another wave only for j >= 256 (through LDS: 10 of the 78 passes).
__device__ __forceinline__ bool nms_before(float a, int ia, float b, int ib) {   // "a precedes b" in the final descending order
    return (a > b) || (a == b && ia > ib) || (b != b && a == a);
}

__global__ __launch_bounds__(1024) void k_nms_sort(const float* __restrict__ scores, int n, int* __restrict__ order, const int* __restrict__ n_dev,
                                                   int n_cap) {
    __shared__ float key[NMS_MAX];
    __shared__ int idx[NMS_MAX];
    n = nms_count(n, n_dev, n_cap);
    const int t = threadIdx.x;
    float kv[4];
    int iv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * t + r;
        kv[r] = i < n ? scores[i] : -__builtin_inff();   // padding (-inf, -1) sinks to the end
        iv[r] = i < n ? i : -1;
    }
    // the network only has to order the first P positions, P = the power of two that holds the n candidates (the padding behind them is (-inf, -1)
    // everywhere and never written out): 45 passes and one LDS exchange for 300 candidates instead of 78 and ten
    int P = 4;
    while (P < n) P <<= 1;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = r ^ j;
                    if (q > r) {
                        const int i = 4 * t + r;
                        const bool desc = (i & k) == 0;
                        const bool a_first = nms_before(kv[r], iv[r], kv[q], iv[q]);
                        if (desc ? !a_first : a_first) {
                            const float tk = kv[r]; kv[r] = kv[q]; kv[q] = tk;
                            const int ti = iv[r]; iv[r] = iv[q]; iv[q] = ti;
                        }
                    }
                }
            } else if (j < 256) {
                const int lj = j >> 2;   // lane distance
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ok = __shfl_xor(kv[r], lj);
                    const int oi = __shfl_xor(iv[r], lj);
                    const int i = 4 * t + r;
                    const bool lower = (i & j) == 0;                 // this thread holds the lower position of the pair
                    const bool desc = (i & k) == 0;
                    const bool mine_first = nms_before(kv[r], iv[r], ok, oi);
                    // the lower position keeps the element that precedes (descending block) / follows (ascending block)
                    const bool keep_mine = (lower == desc) ? mine_first : !mine_first;
                    if (!keep_mine) { kv[r] = ok; iv[r] = oi; }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) { key[4 * t + r] = kv[r]; idx[4 * t + r] = iv[r]; }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * t + r, p = i ^ j;
                    const float ok = key[p];
                    const int oi = idx[p];
                    const bool lower = (i & j) == 0;
                    const bool desc = (i & k) == 0;
                    const bool mine_first = nms_before(kv[r], iv[r], ok, oi);
                    const bool keep_mine = (lower == desc) ? mine_first : !mine_first;
                    if (!keep_mine) { kv[r] = ok; iv[r] = oi; }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * t + r < n) order[4 * t + r] = iv[r];
}

__device__ __forceinline__ float box_volume(const float* b) { return (b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]); }

__global__ __launch_bounds__(64) void k_nms_mask(const float* __restrict__ boxes, const int64_t* __restrict__ classes,
                                                 const int* __restrict__ order, int n, float thresh,
                                                 unsigned long long* __restrict__ mask, const int* __restrict__ n_dev, int n_cap) {
    // block (bx, by): rows 64*by.., columns 64*bx.. of the sorted list
    n = nms_count(n, n_dev, n_cap);
    const int words = (n + 63) / 64;
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb || cb >= words) return;  // only j > i matters; the grid is sized for n_cap
    __shared__ float cbox[64][6];
    __shared__ long long ccls[64];
    const int cj = cb * 64 + threadIdx.x;
    if (cj < n) {
        const int o = order[cj];
#pragma unroll
        for (int k = 0; k < 6; ++k) cbox[threadIdx.x][k] = boxes[o * 6 + k];
        ccls[threadIdx.x] = classes[o];
    }
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    const int oi = order[i];
    float bi[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) bi[k] = boxes[oi * 6 + k];
    const long long ci = classes[oi];
    const float ai = box_volume(bi);
    unsigned long long bits = 0ull;
    const int jn = min(64, n - cb * 64);
    for (int t = 0; t < jn; ++t) {
        const int j = cb * 64 + t;
        if (j <= i) continue;
        const float* bj = cbox[t];
        const float l = fmaxf(0.0f, fminf(bi[3], bj[3]) - fmaxf(bi[0], bj[0]));
        const float w = fmaxf(0.0f, fminf(bi[4], bj[4]) - fmaxf(bi[1], bj[1]));
        const float h = fmaxf(0.0f, fminf(bi[5], bj[5]) - fmaxf(bi[2], bj[2]));
        const float inter = l * w * h;
        float iou = inter / (ai + box_volume(bj) - inter);
        iou = iou * (ci == ccls[t] ? 1.0f : 0.0f);
        if (!(iou <= thresh)) bits |= 1ull << t;
    }
    mask[(int64_t)i * words + cb] = bits;
}

// 64 candidates at a time.  Inside a block the greedy walk only needs the block's own 64 x 64 corner of the bit matrix
// (one word per candidate, one candidate per lane): a scalar loop over the still-alive bits -- find-first-set, read the
// winner's word with v_readlane, clear what it suppresses -- with no memory access in the chain.  Only the rows of the
// block's winners are then fetched (a few per block, issued together) and ORed into the removed-set of the later words.
__device__ __forceinline__ unsigned long long nms_readlane64(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void k_nms_sweep(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                  int n, int64_t* __restrict__ keep, int64_t* __restrict__ n_keep, const int* __restrict__ n_dev,
                                                  int n_cap) {
    const int lane = threadIdx.x;
    n = nms_count(n, n_dev, n_cap);
    const int words = (n + 63) / 64;  // <= 64
    unsigned long long removed = 0ull;  // lane w holds bits [64w, 64w+64)
    int kept = 0;
    // this block's corner word and original index per candidate, fetched one block ahead
    unsigned long long diag = lane < n ? mask[(int64_t)lane * words] : 0ull;
    int ordv = lane < n ? order[lane] : 0;
    for (int b = 0; b < words; ++b) {
        const int base = b * 64;
        const int nb = min(64, n - base);
        const unsigned long long cur_diag = diag;
        const int cur_ord = ordv;
        {   // prefetch the next block's corner
            const int c = base + 64 + lane;
            diag = c < n ? mask[(int64_t)c * words + (b + 1)] : 0ull;
            ordv = c < n ? order[c] : 0;
        }
        const unsigned long long valid = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
        unsigned long long alive = ~nms_readlane64(removed, b) & valid;   // wave-uniform
        unsigned long long winners = 0ull;
        while (alive) {
            const int i = __builtin_ctzll(alive);
            const unsigned long long bit = 1ull << i;
            winners |= bit;
            alive &= ~bit;
            alive &= ~nms_readlane64(cur_diag, i);
        }
        // keep[] in walk order: lane i's slot is the number of winners before it
        if ((winners >> lane) & 1ull) keep[kept + __builtin_popcountll(winners & ((1ull << lane) - 1ull))] = (int64_t)cur_ord;
        kept += __builtin_popcountll(winners);
        // rows of the winners -> removed-set of the later blocks (words left of the diagonal were never written)
        const bool later = lane > b && lane < words;
        unsigned long long w = winners;
        while (w) {      // sixteen winners' rows in flight (four at a time paid one memory latency per four winners: ~40 winners a block)
            unsigned long long r[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                r[k] = 0ull;
                if (w) {
                    const int i = __builtin_ctzll(w);
                    w &= w - 1ull;
                    if (later) r[k] = mask[(int64_t)(base + i) * words + lane];
                }
            }
            unsigned long long acc = 0ull;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc |= r[k];
            removed |= acc;
        }
    }
    if (lane == 0) *n_keep = kept;
}

extern "C" int64_t ndet_nms_workspace_bytes(int n) {
    if (n <= 0) return 0;
    const size_t words = (size_t)(n + 63) / 64;
    return (int64_t)((size_t)n * words * 8 + (size_t)n * 4 + 64);
}

extern "C" int ndet_aligned_3d_nms(const float* boxes, const float* scores, const int64_t* classes, int n, float thresh,
                                   int64_t* keep, int64_t* n_keep, void* workspace, void* stream) {
    const char* fn = "ndet_aligned_3d_nms";
    NDET_REQUIRE(n_keep && (n == 0 || (boxes && scores && classes && keep && workspace)), NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n >= 0, NDET_E_INVALID, "%s: negative n", fn);
    NDET_REQUIRE(n <= NMS_MAX, NDET_E_UNSUPPORTED, "%s: n=%d candidates, at most %d supported", fn, n, NMS_MAX);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(n_keep, 0, sizeof(int64_t), st);
        NDET_REQUIRE(e == hipSuccess, NDET_E_LAUNCH, "%s: memset failed", fn);
        return NDET_OK;
    }
    const int words = (n + 63) / 64;
    unsigned long long* mask = (unsigned long long*)workspace;
    int* order = (int*)((char*)workspace + (size_t)n * words * 8);
    hipLaunchKernelGGL(k_nms_sort, dim3(1), dim3(1024), 0, st, scores, n, order, (const int*)nullptr, 0);
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words), dim3(64), 0, st, boxes, classes, order, n, thresh, mask, (const int*)nullptr, 0);
    hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), 0, st, mask, order, n, keep, n_keep, (const int*)nullptr, 0);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// picked candidates -> one host-bound buffer: header {n_keep, n_candidates, status, 0} + rows [x, y, z_bottom, dx, dy, dz, yaw = 0, score, label]
// (the DepthInstance3DBoxes layout of imvoxel_head_v2.py:546-555 + the bbox3d2result triple).  status bit 0: more candidates than n_cap;
// bit 1: a level holds more than nms_pre survivors (its top-k cut would apply); bit 2: more picks than rows.
__global__ __launch_bounds__(256) void k_pack_detections(const int64_t* __restrict__ keep, const int64_t* __restrict__ n_keep,
                                                         const int* __restrict__ counts, int n_levels, int nms_pre, int n_cap, int k_cap,
                                                         const float* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int64_t* __restrict__ labels, float* __restrict__ out,
                                                         const unsigned* __restrict__ range_guard) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = counts[n_levels];
    const int k = n > n_cap ? 0 : (int)*n_keep;
    if (i == 0) {
        int status = n > n_cap ? 1 : 0;
        if (nms_pre > 0)
            for (int l = 0; l < n_levels; ++l)
                if (counts[l] > nms_pre) status |= 2;
        if (k > k_cap) status |= 4;
        // header word 3: the scene's range-guard word (conv_common.hpp::conv_guard_check) rides to the host with the picks -- no extra copy, no sync
        out[0] = (float)k; out[1] = (float)n; out[2] = (float)status; out[3] = range_guard ? (float)(*range_guard & 1u) : 0.0f;
    }
    if (i >= k || i >= k_cap) return;
    const int64_t c = keep[i];
    const float* b = boxes + c * 6;
    float* o = out + 4 + (int64_t)i * 9;
    const float dz = b[5] - b[2];
    o[0] = (b[0] + b[3]) / 2.0f; o[1] = (b[1] + b[4]) / 2.0f;
    o[2] = (b[2] + b[5]) / 2.0f + dz * -0.5f;          // gravity centre -> bottom centre, as the box container re-bases it
    o[3] = b[3] - b[0]; o[4] = b[4] - b[1]; o[5] = dz; o[6] = 0.0f;
    o[7] = scores[c];
    o[8] = (float)labels[c];
}

extern "C" int ndet_nms_pack_detections(const float* cand_boxes, const float* cand_scores, const int64_t* cand_labels, const int* counts,
                                        int n_levels, int nms_pre, int n_cap, float thresh, int64_t* keep, int64_t* n_keep, void* workspace,
                                        float* out_packed, int k_cap, const unsigned* range_guard, void* stream) {
    const char* fn = "ndet_nms_pack_detections";
    NDET_REQUIRE(cand_boxes && cand_scores && cand_labels && counts && keep && n_keep && workspace && out_packed, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_levels >= 1 && n_levels <= 4 && n_cap >= 1 && n_cap <= NMS_MAX && k_cap >= 1, NDET_E_INVALID, "%s: bad sizes", fn);
    hipStream_t st = (hipStream_t)stream;
    const int words = (n_cap + 63) / 64;
    unsigned long long* mask = (unsigned long long*)workspace;
    int* order = (int*)((char*)workspace + (size_t)n_cap * words * 8);
    const int* n_dev = counts + n_levels;
    hipLaunchKernelGGL(k_nms_sort, dim3(1), dim3(1024), 0, st, cand_scores, 0, order, n_dev, n_cap);
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words), dim3(64), 0, st, cand_boxes, cand_labels, order, 0, thresh, mask, n_dev, n_cap);
    hipLaunchKernelGGL(k_nms_sweep, dim3(1), dim3(64), 0, st, mask, order, 0, keep, n_keep, n_dev, n_cap);
    hipLaunchKernelGGL(k_pack_detections, dim3((k_cap + 255) / 256), dim3(256), 0, st, keep, n_keep, counts, n_levels, nms_pre, n_cap, k_cap, cand_boxes,
                       cand_scores, cand_labels, out_packed, range_guard);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// A14 decode: one pass over a head level's fused conv output (N, 1 + n_reg + n_cls) = [centerness | reg | cls logits]:
//   score_k = sigmoid(cls_k) * sigmoid(centerness) * valid;  best = max_k, label = first argmax          (imvoxel_head_v2.py:266-271)
//   d = exp(scale * reg);  box = (p - d0, p - d2, p - d4, p + d1, p + d3, p + d5) at the voxel lower corner p   (:447,:547-555)
// with p = idx * voxel_size + new_origin computed as get_points does.  Replaces ~25 elementwise launches per level.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_decode(const float* __restrict__ raw, int n_cls, const uint8_t* __restrict__ valid,
                                                     const float* __restrict__ scale, int nx, int ny, int nz, float vx, float vy, float vz,
                                                     float ox, float oy, float oz, float* __restrict__ best, int64_t* __restrict__ label,
                                                     float* __restrict__ boxes) {
    const int N = nx * ny * nz;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int stride = 7 + n_cls;
    const float* r = raw + (int64_t)n * stride;
    const float ctr = 1.0f / (1.0f + expf(-r[0]));
    const float v = valid[n] ? 1.0f : 0.0f;
    float bs = -1.0f;
    int bl = 0;
    for (int k = 0; k < n_cls; ++k) {
        const float sc = ((1.0f / (1.0f + expf(-r[7 + k]))) * ctr) * v;
        if (sc > bs) { bs = sc; bl = k; }
    }
    best[n] = bs;
    label[n] = bl;
    const int iz = n % nz, iy = (n / nz) % ny, ix = n / (nz * ny);
    const float px = (float)ix * vx + ox, py = (float)iy * vy + oy, pz = (float)iz * vz + oz;
    const float s = scale[0];
    float d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = expf(r[1 + k] * s);
    float* b = boxes + (int64_t)n * 6;
    b[0] = px - d[0]; b[1] = py - d[2]; b[2] = pz - d[4];
    b[3] = px + d[1]; b[4] = py + d[3]; b[5] = pz + d[5];
}

extern "C" int ndet_head_decode(const float* raw, int n_cls, const uint8_t* valid, const float* scale, int nx, int ny, int nz,
                                const float* voxel_size_host, const float* origin_host, float* best, int64_t* label, float* boxes, void* stream) {
    const char* fn = "ndet_head_decode";
    NDET_REQUIRE(raw && valid && scale && voxel_size_host && origin_host && best && label && boxes, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(nx > 0 && ny > 0 && nz > 0 && n_cls > 0, NDET_E_INVALID, "%s: sizes must be positive", fn);
    volatile float hx = (float)nx / 2.0f, hy = (float)ny / 2.0f, hz = (float)nz / 2.0f;
    volatile float mx = hx * voxel_size_host[0], my = hy * voxel_size_host[1], mz = hz * voxel_size_host[2];
    const float ox = origin_host[0] - mx, oy = origin_host[1] - my, oz = origin_host[2] - mz;
    const int N = nx * ny * nz;
    hipLaunchKernelGGL(k_head_decode, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, n_cls, valid, scale, nx, ny, nz,
                       voxel_size_host[0], voxel_size_host[1], voxel_size_host[2], ox, oy, oz, best, label, boxes);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// Every level's validity mask and decode in ONE launch (the one-scene loop's tail is a chain of tiny dependent kernels: six launches of ~5 us with
// their gaps for what is 29 200 independent voxels at the configs' sizes).  Same arithmetic as k_level_valid + k_head_decode, level by level.
struct DecodeLevels {
    const float* raw[4];       // (X_l, Y_l, Z_l, 7 + n_cls) head outputs
    const float* scale[4];     // the level's learnable Scale (device scalar)
    float* best[4];
    int64_t* label[4];
    float* boxes[4];
    int nx[4], ny[4], nz[4], f[4];     // level grid, integer down-scale against the full-resolution validity volume
    float vx[4], vy[4], vz[4], ox[4], oy[4], oz[4];
    int start[5];              // first global thread of each level (prefix sums of the voxel counts)
    int n_levels, n_cls, X, Y, Z;
    const float* valid;        // (X, Y, Z) floats
};

__global__ __launch_bounds__(256) void k_head_decode_levels(const DecodeLevels L) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L.start[L.n_levels]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < L.n_levels && t >= L.start[k]) l = k;
    const int n = t - L.start[l];
    const int nx = L.nx[l], ny = L.ny[l], nz = L.nz[l], f = L.f[l];
    const int iz = n % nz, iy = (n / nz) % ny, ix = n / (nz * ny);
    // ---- k_level_valid ----
    float vsum;
    if (f == 1) vsum = L.valid[n];
    else {
        const int x0 = f * ix + f / 2 - 1, y0 = f * iy + f / 2 - 1, z0 = f * iz + f / 2 - 1;
        vsum = 0.0f;
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dz = 0; dz < 2; ++dz) vsum += 0.125f * L.valid[((int64_t)(x0 + dx) * L.Y + (y0 + dy)) * L.Z + (z0 + dz)];
    }
    const float v = rintf(vsum) != 0.0f ? 1.0f : 0.0f;
    // ---- k_head_decode ----
    const int stride = 7 + L.n_cls;
    const float* r = L.raw[l] + (int64_t)n * stride;
    const float ctr = 1.0f / (1.0f + expf(-r[0]));
    float bs = -1.0f;
    int bl = 0;
    for (int k = 0; k < L.n_cls; ++k) {
        const float sc = ((1.0f / (1.0f + expf(-r[7 + k]))) * ctr) * v;
        if (sc > bs) { bs = sc; bl = k; }
    }
    L.best[l][n] = bs;
    L.label[l][n] = bl;
    const float px = (float)ix * L.vx[l] + L.ox[l], py = (float)iy * L.vy[l] + L.oy[l], pz = (float)iz * L.vz[l] + L.oz[l];
    const float s = L.scale[l][0];
    float d[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = expf(r[1 + k] * s);
    float* b = L.boxes[l] + (int64_t)n * 6;
    b[0] = px - d[0]; b[1] = py - d[2]; b[2] = pz - d[4];
    b[3] = px + d[1]; b[4] = py + d[3]; b[5] = pz + d[5];
}

extern "C" int ndet_head_decode_levels(int n_levels, const float* const* raw_host, const float* const* scale_host, const int* dims_host,
                                       const int* factor_host, const float* voxel_size_host, const float* origin_host, int n_cls, const float* valid,
                                       int X, int Y, int Z, float* const* best_host, int64_t* const* label_host, float* const* boxes_host, void* stream) {
    const char* fn = "ndet_head_decode_levels";
    NDET_REQUIRE(raw_host && scale_host && dims_host && factor_host && voxel_size_host && origin_host && valid && best_host && label_host && boxes_host,
                 NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_levels >= 1 && n_levels <= 4 && n_cls > 0 && X > 0 && Y > 0 && Z > 0, NDET_E_INVALID, "%s: 1..4 levels, positive sizes", fn);
    DecodeLevels L;
    L.n_levels = n_levels; L.n_cls = n_cls; L.X = X; L.Y = Y; L.Z = Z; L.valid = valid;
    int64_t total = 0;
    for (int l = 0; l < 4; ++l) {
        if (l >= n_levels) {
            L.raw[l] = nullptr; L.scale[l] = nullptr; L.best[l] = nullptr; L.label[l] = nullptr; L.boxes[l] = nullptr;
            L.nx[l] = L.ny[l] = L.nz[l] = L.f[l] = 1; L.vx[l] = L.vy[l] = L.vz[l] = L.ox[l] = L.oy[l] = L.oz[l] = 0.0f;
            continue;
        }
        const int nx = dims_host[3 * l], ny = dims_host[3 * l + 1], nz = dims_host[3 * l + 2], f = factor_host[l];
        NDET_REQUIRE(raw_host[l] && scale_host[l] && best_host[l] && label_host[l] && boxes_host[l], NDET_E_INVALID, "%s: null pointer at level %d", fn, l);
        NDET_REQUIRE(nx > 0 && ny > 0 && nz > 0 && f >= 1 && (f == 1 || f % 2 == 0) && nx * f == X && ny * f == Y && nz * f == Z, NDET_E_UNSUPPORTED,
                     "%s: level %d (%d x %d x %d, factor %d) is not an integer (1 or even) down-scale of the %d x %d x %d validity volume", fn, l, nx, ny, nz, f, X, Y, Z);
        L.raw[l] = raw_host[l]; L.scale[l] = scale_host[l]; L.best[l] = best_host[l]; L.label[l] = label_host[l]; L.boxes[l] = boxes_host[l];
        L.nx[l] = nx; L.ny[l] = ny; L.nz[l] = nz; L.f[l] = f;
        // the level's voxel pitch and the grid's corner, evaluated as ndet_head_decode does (volatile: no contraction of the host arithmetic)
        const float sx = voxel_size_host[3 * l], sy = voxel_size_host[3 * l + 1], sz = voxel_size_host[3 * l + 2];
        volatile float hx = (float)nx / 2.0f, hy = (float)ny / 2.0f, hz = (float)nz / 2.0f;
        volatile float mx = hx * sx, my = hy * sy, mz = hz * sz;
        L.vx[l] = sx; L.vy[l] = sy; L.vz[l] = sz;
        L.ox[l] = origin_host[0] - mx; L.oy[l] = origin_host[1] - my; L.oz[l] = origin_host[2] - mz;
        L.start[l] = (int)total;
        total += (int64_t)nx * ny * nz;
        NDET_REQUIRE(total < ((int64_t)1 << 31), NDET_E_UNSUPPORTED, "%s: too many voxels", fn);
    }
    for (int l = n_levels; l <= 4; ++l) L.start[l] = (int)total;
    hipLaunchKernelGGL(k_head_decode_levels, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// ------------------------------------------------------------------------------------------------
// Head post-processing between the decode and the NMS in two launches instead of ~60 library kernels.
//
// k_level_valid: `F.interpolate(valid, size, mode='trilinear').round().bool()` of imvoxel_head_v2.py:442-449 for the
// integer down-scales the FPN levels have (1, 2, 4, ...): with align_corners=False an even factor f samples exactly half
// way between input voxels f i + f/2 - 1 and f i + f/2 on every axis, so the value is the mean of that 2x2x2 block
// (weights 0.125, exact in fp32 for view counts); round-half-even then bool = "mean > 0.5".
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_level_valid(const float* __restrict__ valid, int X, int Y, int Z, int f, uint8_t* __restrict__ out) {
    const int nx = X / f, ny = Y / f, nz = Z / f;
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nx * ny * nz) return;
    const int iz = n % nz, iy = (n / nz) % ny, ix = n / (nz * ny);
    if (f == 1) {
        out[n] = rintf(valid[n]) != 0.0f;
        return;
    }
    const int x0 = f * ix + f / 2 - 1, y0 = f * iy + f / 2 - 1, z0 = f * iz + f / 2 - 1;
    float s = 0.0f;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) s += 0.125f * valid[((int64_t)(x0 + dx) * Y + (y0 + dy)) * Z + (z0 + dz)];
    out[n] = rintf(s) != 0.0f;
}

extern "C" int ndet_level_valid(const float* valid, int X, int Y, int Z, int factor, uint8_t* out, void* stream) {
    const char* fn = "ndet_level_valid";
    NDET_REQUIRE(valid && out, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(X > 0 && Y > 0 && Z > 0 && factor >= 1, NDET_E_INVALID, "%s: sizes must be positive", fn);
    NDET_REQUIRE(factor == 1 || (factor % 2 == 0 && X % factor == 0 && Y % factor == 0 && Z % factor == 0), NDET_E_UNSUPPORTED,
                 "%s: factor %d must be 1 or an even divisor of the grid", fn, factor);
    const int n = (X / factor) * (Y / factor) * (Z / factor);
    hipLaunchKernelGGL(k_level_valid, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, valid, X, Y, Z, factor, out);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// k_select_candidates: `scores > score_thr` over the concatenated levels (imvoxel_head_v2.py:533-545), order-preserving
// compaction by one workgroup (ballot + wave prefix + LDS), so the candidate list -- and with it every tie-break of the
// NMS -- is reproducible.  counts[l] = survivors of level l, counts[n_levels] = total.
struct SelectLevels {
    const float* best[4];
    const int64_t* label[4];
    const float* boxes[4];
    int n[4];
    int n_levels;
};

__global__ __launch_bounds__(1024) void k_select_candidates(const SelectLevels lv, float thr, float* __restrict__ o_best, int64_t* __restrict__ o_label,
                                                            float* __restrict__ o_boxes, int* __restrict__ counts) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int l = 0; l < lv.n_levels; ++l) {
        const int level_base = base_s;
        for (int i0 = 0; i0 < lv.n[l]; i0 += 1024) {
            const int i = i0 + tid;
            const bool keep = i < lv.n[l] && lv.best[l][i] > thr;
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int off = base_s;
            for (int w = 0; w < wave; ++w) off += wave_cnt[w];
            if (keep) {
                const int o = off + __popcll(m & ((1ull << lane) - 1ull));
                o_best[o] = lv.best[l][i];
                o_label[o] = lv.label[l][i];
#pragma unroll
                for (int k = 0; k < 6; ++k) o_boxes[(int64_t)o * 6 + k] = lv.boxes[l][(int64_t)i * 6 + k];
            }
            __syncthreads();
            if (tid == 0) {
                int t = 0;
                for (int w = 0; w < 16; ++w) t += wave_cnt[w];
                base_s += t;
            }
            __syncthreads();
        }
        if (tid == 0) counts[l] = base_s - level_base;
    }
    if (tid == 0) counts[lv.n_levels] = base_s;
}

// The same compaction with the per-level ``topk(nms_pre)`` cut of imvoxel_head_v2.py:272-276 decided ON THE DEVICE: when more than
// nms_pre voxels of a level clear the threshold, the nms_pre-th largest score is found by a 3-round radix select over the float
// bits (11 + 11 + 10 bits, a 2048-bin LDS histogram per round; positive floats order like their bit patterns) and only scores
// above it -- plus as many equal to it as are needed, first in voxel order -- are kept.  The kept SET is the reference's
// (its order inside a level differs: voxel order here, descending score there; the NMS sorts by score anyway).
// counts[l] = candidates of level l after the cut, counts[n_levels] = total, counts[n_levels + 1] = survivors before any cut.
__global__ __launch_bounds__(1024) void k_select_candidates_topk(const SelectLevels lv, float thr, int nms_pre, float* __restrict__ o_best,
                                                                 int64_t* __restrict__ o_label, float* __restrict__ o_boxes, int* __restrict__ counts,
                                                                 int lds_cap) {
    // a level's scores are walked five times (threshold count, three radix rounds, compaction) by ONE workgroup: each walk from global
    // memory is a chain of dependent-latency loads (25 per thread at 25 600 voxels).  The first walk parks them in LDS (16-byte loads, up
    // to lds_cap floats); levels that do not fit keep reading global memory.
    extern __shared__ __attribute__((aligned(16))) float lsc[];
    __shared__ int wave_cnt[16], wave_eq[16];
    __shared__ int base_s, eq_s, raw_s, sel_need;
    __shared__ unsigned sel_prefix;
    __shared__ int hist[2048];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { base_s = 0; raw_s = 0; }
    __syncthreads();
    for (int l = 0; l < lv.n_levels; ++l) {
        const int n_l = lv.n[l];
        const float* __restrict__ gbest = lv.best[l];
        const bool cached = n_l <= lds_cap && (((uintptr_t)gbest) & 15) == 0;
        auto best_at = [&](int i) -> float { return cached ? lsc[i] : gbest[i]; };
        // ---- survivors of the threshold ----
        int c = 0;
        if (cached) {
            __syncthreads();                               // the previous level's readers are done with lsc
            const int n4 = n_l >> 2;
            for (int i = tid; i < n4; i += 1024) {
                const float4 v = reinterpret_cast<const float4*>(gbest)[i];
                reinterpret_cast<float4*>(lsc)[i] = v;
                c += (v.x > thr ? 1 : 0) + (v.y > thr ? 1 : 0) + (v.z > thr ? 1 : 0) + (v.w > thr ? 1 : 0);
            }
            for (int i = 4 * n4 + tid; i < n_l; i += 1024) {
                const float v = gbest[i];
                lsc[i] = v;
                c += v > thr ? 1 : 0;
            }
        } else {
            for (int i = tid; i < n_l; i += 1024) c += gbest[i] > thr ? 1 : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0) wave_cnt[wave] = c;
        __syncthreads();
        c = 0;
        for (int w2 = 0; w2 < 16; ++w2) c += wave_cnt[w2];
        __syncthreads();
        // ---- the nms_pre-th largest key, if the cut applies ----
        unsigned K = 0u;      // keep keys > K ...
        int ties_need = 0;    // ... and the first ties_need keys == K
        if (nms_pre > 0 && c > nms_pre) {
            unsigned prefix = 0u;
            int need = nms_pre, bits_done = 0;
            for (int r = 0; r < 3; ++r) {
                const int width = r < 2 ? 11 : 10, shift = 32 - bits_done - width;
                for (int b = tid; b < 2048; b += 1024) hist[b] = 0;
                __syncthreads();
                for (int i = tid; i < n_l; i += 1024) {
                    const float v = best_at(i);
                    if (v > thr) {
                        const unsigned key = __float_as_uint(v);
                        if (bits_done == 0 || (key >> (32 - bits_done)) == prefix) atomicAdd(&hist[(key >> shift) & ((1u << width) - 1u)], 1);
                    }
                }
                __syncthreads();
                if (wave == 0) {   // from the top bin down: the bin where the count of larger keys crosses `need`
                    const int per = (1 << width) / 64;
                    int chunk = 0;
                    for (int b = 0; b < per; ++b) chunk += hist[lane * per + b];
                    int suf = chunk;   // sum over lanes >= this one
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int t = __shfl_down(suf, off);
                        if (lane + off < 64) suf += t;
                    }
                    const unsigned long long ok = __ballot(suf >= need);
                    const int L = 63 - __builtin_clzll(ok);
                    if (lane == L) {
                        int cum = suf - chunk, bin = lane * per;
                        for (int b = per - 1; b >= 0; --b) {
                            const int hb = hist[lane * per + b];
                            if (cum + hb >= need) { bin = lane * per + b; break; }
                            cum += hb;
                        }
                        sel_prefix = (prefix << width) | (unsigned)bin;
                        sel_need = need - cum;
                    }
                }
                __syncthreads();
                prefix = sel_prefix;
                need = sel_need;
                bits_done += width;
                __syncthreads();
            }
            K = prefix;
            ties_need = need;
        }
        // ---- ordered compaction: every thread owns a contiguous run of voxels; two block-wide scans (ties, then kept) give its offsets ----
        const int level_base = base_s;
        const int per = (n_l + 1023) / 1024;
        const int i_lo = min(tid * per, n_l), i_hi = min(i_lo + per, n_l);
        int n_gt = 0, n_eq = 0;
        for (int i = i_lo; i < i_hi; ++i) {
            const float v = best_at(i);
            if (v > thr) {
                const unsigned key = __float_as_uint(v);
                n_gt += key > K ? 1 : 0;
                n_eq += (K != 0u && key == K) ? 1 : 0;
            }
        }
        auto block_exclusive = [&](int x, int* scratch, int& total) {   // exclusive prefix of x over the 1024 threads, in thread order
            int inc = x;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off);
                if (lane >= off) inc += t;
            }
            if (lane == 63) scratch[wave] = inc;
            __syncthreads();
            int before = 0, all = 0;
            for (int w2 = 0; w2 < 16; ++w2) {
                const int t = scratch[w2];
                before += w2 < wave ? t : 0;
                all += t;
            }
            __syncthreads();
            total = all;
            return before + inc - x;
        };
        int tot_eq, tot_keep;
        const int eq_before = block_exclusive(n_eq, wave_eq, tot_eq);
        const int eq_keep = max(0, min(n_eq, ties_need - eq_before));      // ties kept in this run: the first ones in voxel order
        const int off0 = block_exclusive(n_gt + eq_keep, wave_cnt, tot_keep);
        int o = level_base + off0, eq_left = eq_keep;
        for (int i = i_lo; i < i_hi; ++i) {
            const float v = best_at(i);
            if (!(v > thr)) continue;
            const unsigned key = __float_as_uint(v);
            bool keep = key > K;
            if (!keep && K != 0u && key == K && eq_left > 0) { keep = true; --eq_left; }
            if (keep) {
                o_best[o] = v;
                o_label[o] = lv.label[l][i];
#pragma unroll
                for (int k = 0; k < 6; ++k) o_boxes[(int64_t)o * 6 + k] = lv.boxes[l][(int64_t)i * 6 + k];
                ++o;
            }
        }
        if (tid == 0) { base_s = level_base + tot_keep; raw_s += c; }
        __syncthreads();
        if (tid == 0) counts[l] = base_s - level_base;
    }
    if (tid == 0) { counts[lv.n_levels] = base_s; counts[lv.n_levels + 1] = raw_s; }
}

extern "C" int ndet_select_candidates_topk(int n_levels, const float* const* best, const int64_t* const* label, const float* const* boxes,
                                           const int* n, float score_thr, int nms_pre, float* out_best, int64_t* out_label, float* out_boxes,
                                           int* counts, void* stream) {
    const char* fn = "ndet_select_candidates_topk";
    NDET_REQUIRE(best && label && boxes && n && out_best && out_label && out_boxes && counts, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_levels >= 1 && n_levels <= 4, NDET_E_UNSUPPORTED, "%s: 1..4 levels", fn);
    NDET_REQUIRE(score_thr >= 0.0f, NDET_E_UNSUPPORTED, "%s: the radix select orders positive scores by their bit patterns: score_thr must be >= 0", fn);
    SelectLevels lv;
    lv.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        NDET_REQUIRE(best[l] && label[l] && boxes[l] && n[l] > 0, NDET_E_INVALID, "%s: level %d: null pointer / empty", fn, l);
        lv.best[l] = best[l]; lv.label[l] = label[l]; lv.boxes[l] = boxes[l]; lv.n[l] = n[l];
    }
    // LDS for the scores of the largest level that fits (36 000 floats = 141 KB beside the 8 KB histogram); larger levels read global memory
    constexpr int LDS_CAP = 36000;
    int cap = 0;
    for (int l = 0; l < n_levels; ++l)
        if (n[l] <= LDS_CAP && n[l] > cap) cap = n[l];
    // the LDS limit is a per-device attribute of the kernel: raised once per device; where it cannot be raised (a device with 64 KB of LDS) the
    // kernel takes every level's scores from global memory (cap = 0) instead of failing
    static int attr_state[16] = {0};    // per device: 0 unknown, 1 raised, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (attr_state[dev] == 0) {
        hipError_t e = hipFuncSetAttribute((const void*)k_select_candidates_topk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_CAP * sizeof(float)));
        if (e != hipSuccess) (void)hipGetLastError();
        attr_state[dev] = e == hipSuccess ? 1 : -1;
    }
    if (attr_state[dev] < 0 && (size_t)((cap + 3) / 4 * 4) * sizeof(float) > 48 * 1024) cap = 0;
    const size_t lds = (size_t)((cap + 3) / 4 * 4) * sizeof(float);
    hipLaunchKernelGGL(k_select_candidates_topk, dim3(1), dim3(1024), lds, (hipStream_t)stream, lv, score_thr, nms_pre, out_best, out_label, out_boxes,
                       counts, cap);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

extern "C" int ndet_select_candidates(int n_levels, const float* const* best, const int64_t* const* label, const float* const* boxes,
                                      const int* n, float score_thr, float* out_best, int64_t* out_label, float* out_boxes, int* counts,
                                      void* stream) {
    const char* fn = "ndet_select_candidates";
    NDET_REQUIRE(best && label && boxes && n && out_best && out_label && out_boxes && counts, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_levels >= 1 && n_levels <= 4, NDET_E_UNSUPPORTED, "%s: 1..4 levels", fn);
    SelectLevels lv;
    lv.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) {
        NDET_REQUIRE(best[l] && label[l] && boxes[l] && n[l] > 0, NDET_E_INVALID, "%s: level %d: null pointer / empty", fn, l);
        lv.best[l] = best[l]; lv.label[l] = label[l]; lv.boxes[l] = boxes[l]; lv.n[l] = n[l];
    }
    hipLaunchKernelGGL(k_select_candidates, dim3(1), dim3(1024), 0, (hipStream_t)stream, lv, score_thr, out_best, out_label, out_boxes, counts);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}

// picked candidates -> (centre, size) boxes, scores, labels in pick order (imvoxel_head_v2.py:546-555)
__global__ __launch_bounds__(256) void k_gather_detections(const int64_t* __restrict__ keep, int n_keep, const float* __restrict__ boxes,
                                                           const float* __restrict__ scores, const int64_t* __restrict__ labels,
                                                           float* __restrict__ o_boxes, float* __restrict__ o_scores, int64_t* __restrict__ o_labels) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keep) return;
    const int64_t k = keep[i];
    const float* b = boxes + k * 6;
    float* o = o_boxes + (int64_t)i * 6;
    o[0] = (b[0] + b[3]) / 2.0f; o[1] = (b[1] + b[4]) / 2.0f; o[2] = (b[2] + b[5]) / 2.0f;
    o[3] = b[3] - b[0]; o[4] = b[4] - b[1]; o[5] = b[5] - b[2];
    o_scores[i] = scores[k];
    o_labels[i] = labels[k];
}

extern "C" int ndet_gather_detections(const int64_t* keep, int n_keep, const float* boxes, const float* scores, const int64_t* labels,
                                      float* out_boxes, float* out_scores, int64_t* out_labels, void* stream) {
    const char* fn = "ndet_gather_detections";
    NDET_REQUIRE(keep && boxes && scores && labels && out_boxes && out_scores && out_labels, NDET_E_INVALID, "%s: null pointer", fn);
    NDET_REQUIRE(n_keep > 0, NDET_E_INVALID, "%s: n_keep must be positive", fn);
    hipLaunchKernelGGL(k_gather_detections, dim3((n_keep + 255) / 256), dim3(256), 0, (hipStream_t)stream, keep, n_keep, boxes, scores, labels,
                       out_boxes, out_scores, out_labels);
    NDET_CHECK_LAUNCH(fn);
    return NDET_OK;
}
