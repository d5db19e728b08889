// fps.hip — iterative furthest-point sampling for gfx950.
//
// Replaces furthest_point_sampling_kernel (reference: pointnet2/src/sampling_gpu.cu:93-253).
// The reference re-reads xyz and temp from global memory every round and runs an 11-barrier
// shared-memory tree per round.  Here one workgroup owns one cloud and keeps the cloud AND the running
// min-distances in registers for the whole run (N <= 16384); a round is
//     VALU update -> DPP wave max -> DPP wave min over ranks -> one LDS record per wave -> ONE barrier.
// One wavefront per SIMD (256 threads) so the whole CU's VALU works on the cloud and the barrier is cheap.
//
// Tie order.  The reference's winner among equal maxima is fixed by its reduction shape: the strided
// per-thread scan keeps the smallest k (strict '>', sampling_gpu.cu:136-137) and each tree step keeps the
// left operand on ties (__update, :86-91).  Unrolled, that is a total order: among points with the maximal
// value the winner minimises
//       rank(k) = bitrev_{log2 bs}(k % bs) * S + k / bs,   S = ceil(N / bs),
// with bs = min(1024, 2^floor(log2 N)) the reference's block size (cuda_utils.h:10-14).  Points are laid
// out over (lane, register) in RANK order (lane t holds ranks t, t+T, t+2T, ...), so "first maximal register
// of the lane, then minimal rank over lanes" is exactly the reference's winner, and the launch shape is
// free to differ from the reference's.
#include <math.h>
#include <stdlib.h>

#include <mutex>

#include "ogc_common.h"

namespace {

typedef unsigned long long u64;

constexpr int FPS_LDS_XYZ_MAX = 10240; // rank slots whose xyz copy fits in LDS (3 * 4 B * 10240 = 120 KiB)

__device__ __forceinline__ unsigned ogc_wave_min_u32(unsigned v) {
    v = min(v, ogc_dpp_u32<0xB1>(v));
    v = min(v, ogc_dpp_u32<0x4E>(v));
    v = min(v, ogc_dpp_u32<0x141>(v));
    v = min(v, ogc_dpp_u32<0x140>(v));
    const unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}

// rank slot -> point index (>= n for padding slots)
__device__ __forceinline__ int fps_rank_to_k(unsigned rho, int S, int bs_shift) {
    const unsigned rev = rho / (unsigned)S;
    const unsigned kd = rho - rev * (unsigned)S;
    const unsigned tid = bs_shift ? (__brev(rev) >> (32 - bs_shift)) : 0u;
    return (int)((kd << bs_shift) + tid);
}

// Max over the 64 lanes of non-negative floats (or the -1 padding value) compared as signed integers — the bit
// patterns order identically — so every step is ONE v_max_i32 with a DPP operand (fmaxf costs a canonicalising
// v_max + v_max per step, and every dependent VALU op is 8 cycles of latency on the per-round critical path).
__device__ __forceinline__ float fps_wave_max(float v) {
    int x = __float_as_int(v);
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));
    x = max(x, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));
    const int r0 = __builtin_amdgcn_readlane(x, 0), r1 = __builtin_amdgcn_readlane(x, 16);
    const int r2 = __builtin_amdgcn_readlane(x, 32), r3 = __builtin_amdgcn_readlane(x, 48);
    return __int_as_float(max(max(r0, r1), max(r2, r3)));
}

// max of a signed 64-bit key over lanes 0..7 (or 0..15) of a wave (every one of them receives it): quad xor 1, quad xor 2,
// half-row mirror (, row mirror).
// Both halves of the key travel by DPP; one v_cmp_gt_i64 and two v_cndmask per step.
template <int CTRL>
__device__ __forceinline__ long long fps_dpp_max_i64(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xF, 0xF, true);
    const long long o = ((long long)hi << 32) | (long long)(unsigned)lo;
    return o > v ? o : v;
}
// the same step with a row broadcast (row_bcast:15 = 0x142 into rows 1 and 3, row_bcast:31 = 0x143 into rows 2 and 3): lanes
// of the rows that are not written keep their own value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long fps_dpp_bcast_max_i64(long long v) {
    const int vlo = (int)(unsigned)v, vhi = (int)(v >> 32);
    const int lo = __builtin_amdgcn_update_dpp(vlo, vlo, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(vhi, vhi, CTRL, ROW_MASK, 0xF, false);
    const long long o = ((long long)hi << 32) | (long long)(unsigned)lo;
    return o > v ? o : v;
}
// max over all 64 lanes; the lanes of the LAST 16-lane row (48 .. 63) receive it.  No scalar-unit round trip.
__device__ __forceinline__ long long fps_max_to_last_row_i64(long long v) {
    v = fps_dpp_max_i64<0xB1>(v);
    v = fps_dpp_max_i64<0x4E>(v);
    v = fps_dpp_max_i64<0x141>(v);
    v = fps_dpp_max_i64<0x140>(v);
    v = fps_dpp_bcast_max_i64<0x142, 0xA>(v);
    return fps_dpp_bcast_max_i64<0x143, 0xC>(v);
}
// 32-bit versions for max-then-arg without readlanes between the steps: six DPP steps each (the last two row broadcasts), the
// lanes of the last row receive the result
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int fps_dpp_keep(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int fps_max_to_last_row_i32(int v) {
    v = max(v, fps_dpp_keep<0xB1, 0xF>(v));
    v = max(v, fps_dpp_keep<0x4E, 0xF>(v));
    v = max(v, fps_dpp_keep<0x141, 0xF>(v));
    v = max(v, fps_dpp_keep<0x140, 0xF>(v));
    v = max(v, fps_dpp_keep<0x142, 0xA>(v));
    return max(v, fps_dpp_keep<0x143, 0xC>(v));
}
__device__ __forceinline__ unsigned fps_min_to_last_row_u32(unsigned v) {
    v = min(v, (unsigned)fps_dpp_keep<0xB1, 0xF>((int)v));
    v = min(v, (unsigned)fps_dpp_keep<0x4E, 0xF>((int)v));
    v = min(v, (unsigned)fps_dpp_keep<0x141, 0xF>((int)v));
    v = min(v, (unsigned)fps_dpp_keep<0x140, 0xF>((int)v));
    v = min(v, (unsigned)fps_dpp_keep<0x142, 0xA>((int)v));
    return min(v, (unsigned)fps_dpp_keep<0x143, 0xC>((int)v));
}
// (value bits, ~rank) as one signed 64-bit key: larger value first (bit patterns of non-negative floats, and of the -1 padding,
// order as signed integers), then the smaller rank
__device__ __forceinline__ long long fps_key(float v, unsigned rank) {
    return (long long)(((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - rank));
}
template <int LANES> // 4, 8 or 16: max over the first LANES lanes of every 16-lane row (each of them receives its row's)
__device__ __forceinline__ long long fps_max_low_lanes_i64(long long v) {
    v = fps_dpp_max_i64<0xB1>(v);
    v = fps_dpp_max_i64<0x4E>(v);
    if (LANES > 4) v = fps_dpp_max_i64<0x141>(v);
    if (LANES > 8) v = fps_dpp_max_i64<0x140>(v); // row mirror: the other half of the 16-lane row
    return v;
}

// (p - q)^2 summed over the axes for two points at once; per half: ((dx*dx + dy*dy) + dz*dz), one rounding per operation
// (sampling_gpu.cu:133), never contracted.
__device__ __forceinline__ ogc_v2f fps_sqdist2(ogc_v2f px, ogc_v2f py, ogc_v2f pz, ogc_v2f qx, ogc_v2f qy, ogc_v2f qz) {
#pragma clang fp contract(off)
    const ogc_v2f dx = px - qx, dy = py - qy, dz = pz - qz;
    return ogc_sqsum3(dx, dy, dz);
}

template <int PTS, int THREADS, bool LDS_XYZ>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(int n, int m, int bs_shift,
                                                          const float *__restrict__ xyz,
                                                          float *__restrict__ temp, int *__restrict__ idxs,
                                                          const int *__restrict__ ties_in, int *__restrict__ ties_out) {
    extern __shared__ __attribute__((aligned(16))) float fps_smem[];
    // [0..1]: two 64-bit "best of the round" words (double-buffered); then the SoA xyz copy in rank order
    u64 *best_word = reinterpret_cast<u64 *>(fps_smem);
    float *lx = fps_smem + 8;
    const int slots = PTS * THREADS;
    float *ly = lx + slots, *lz = ly + slots;

    const int t = threadIdx.x;
    const int lane = t & (OGC_WAVE - 1);
    const int b = blockIdx.x;
    const float *__restrict__ dataset = xyz + (size_t)b * n * 3;
    float *tmp = temp ? temp + (size_t)b * n : nullptr; // null: the minima start at 1e10 and are not handed back
    int *out = idxs + (size_t)b * m;
    const int S = (n + (1 << bs_shift) - 1) >> bs_shift;
    const int nslot = S << bs_shift; // rank slots in use (>= n)

    // This cloud is the first n samples, in sampling order, of a run whose first m rounds each had a UNIQUE farthest
    // point: sampling it again reproduces that order — round r picks the point that was farthest from the first r
    // samples in the whole cloud, which is sample r itself, and it is still the only maximum — so the answer is
    // 0, 1, ..., m-1 without a single round (see ogc_furthest_point_sampling_chain).
    // L: how many leading samples are known without sampling (the parent run's tie-free rounds)
    const int L = ties_in ? max(1, min(ties_in[b], m)) : 1;
    if (L >= m) {
        for (int r = threadIdx.x; r < m; r += THREADS) out[r] = r;
        if (threadIdx.x == 0 && ties_out) ties_out[b] = ties_in[b];
        return;
    }
    const bool track = ties_out != nullptr;
    int first_tie = 0x7fffffff;

    float px[PTS], py[PTS], pz[PTS], td[PTS];
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const unsigned rho = (unsigned)(t + j * THREADS);
        const int k = rho < (unsigned)nslot ? fps_rank_to_k(rho, S, bs_shift) : n;
        if (k < n) {
            px[j] = dataset[k * 3 + 0];
            py[j] = dataset[k * 3 + 1];
            pz[j] = dataset[k * 3 + 2];
            td[j] = tmp ? tmp[k] : 1e10f;
        } else {
            px[j] = py[j] = pz[j] = 0.0f;
            td[j] = -1.0f; // padding never wins: real values are >= 0 and fminf(d, -1) = -1
        }
        if (LDS_XYZ) {
            lx[rho] = px[j];
            ly[rho] = py[j];
            lz[rho] = pz[j];
        }
    }
    if (t == 0) {
        out[0] = 0; // slot 0 of the output holds point 0; later slots hold RANKS until the final conversion pass
        best_word[0] = 0ull;
        best_word[1] = 0ull;
    }
    // Samples 0 .. L-1 are points 0 .. L-1 (the known prefix): fold samples 0 .. L-2 into the running minima — the same
    // arithmetic as L-2 rounds, but with no maximum to find there is nothing sequential about it — and enter the rounds
    // at r = L with sample L-1 as the point just selected.
    for (int s0 = 0; s0 < L - 1; s0 += 8) { // eight samples' coordinates in flight at a time (wave-uniform loads)
        float sc[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sidx = min(s0 + u, L - 2); // the tail repeats the last sample: taking a minimum twice is harmless
            sc[u][0] = dataset[sidx * 3 + 0]; sc[u][1] = dataset[sidx * 3 + 1]; sc[u][2] = dataset[sidx * 3 + 2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < PTS; ++j)
                td[j] = ogc_min_f32(ogc_sqdist(px[j], py[j], pz[j], sc[u][0], sc[u][1], sc[u][2]), td[j]);
    }
    for (int r = 1 + t; r < L; r += THREADS) out[r] = r; // final indices already (not ranks): skipped by the conversion
    float x1 = dataset[(L - 1) * 3], y1 = dataset[(L - 1) * 3 + 1], z1 = dataset[(L - 1) * 3 + 2];
    __syncthreads();

    unsigned prev_bits = 0u, prev_rho = 0u; // maximum (bit pattern) and winner of the previous round
    float tmax_prev = -1.0f;                // this lane's maximum in the previous round
    auto note_ties = [&](int round) {
        // Did another point attain the previous round's maximum?  Only lanes whose own maximum equals it can hold one:
        // one compare and a ballot per round for every wavefront but the winner's; a lane that holds the value without
        // owning the winning rank slot is a tie, the owner looks for a second register with it.
        const float pm = __uint_as_float(prev_bits);
        const bool cand = tmax_prev == pm;
        if (__builtin_amdgcn_ballot_w64(cand) != 0) { // wave-uniform
            bool tie = false;
            if (cand) {
                if (prev_rho % THREADS != (unsigned)t) tie = true;
                else {
                    int c = 0;
#pragma unroll
                    for (int j = 0; j < PTS; ++j) c += td[j] == pm ? 1 : 0;
                    tie = c > 1;
                }
            }
            if (tie && first_tie > round) first_tie = round;
        }
    };
    for (int r = L; r < m; ++r) {
        const int par = r & 1;
        if (track && r > L) note_ties(r - 1);
        float tmax = -1.0f;
        if constexpr (PTS >= 2) {
            // two of the lane's points per packed instruction (v_pk_add / v_pk_mul_f32): each half is the reference's
            // fp32 expression, un-fused, so the eight scalar operations per point become four
            const ogc_v2f qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
#pragma unroll
            for (int j = 0; j < PTS; j += 2) {
                const ogc_v2f d = fps_sqdist2((ogc_v2f){px[j], px[j + 1]}, (ogc_v2f){py[j], py[j + 1]},
                                              (ogc_v2f){pz[j], pz[j + 1]}, qx, qy, qz);
                td[j] = ogc_min_f32(d.x, td[j]);
                td[j + 1] = ogc_min_f32(d.y, td[j + 1]);
                tmax = fmaxf(tmax, td[j]);
                tmax = fmaxf(tmax, td[j + 1]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PTS; ++j) {
                const float d = ogc_sqdist(px[j], py[j], pz[j], x1, y1, z1);
                td[j] = ogc_min_f32(d, td[j]);
                tmax = fmaxf(tmax, td[j]);
            }
        }
        // lane-local "first register holding the lane maximum": independent of the wave reduction below, so the two
        // dependency chains overlap
        unsigned rho = (unsigned)(t + (PTS - 1) * THREADS);
#pragma unroll
        for (int j = PTS - 2; j >= 0; --j) rho = td[j] == tmax ? (unsigned)(t + j * THREADS) : rho;
        const float wmax = fps_wave_max(tmax);
        const unsigned wrho = ogc_wave_min_u32(tmax == wmax ? rho : 0xFFFFFFFFu);
        // cross-wave: one 64-bit LDS max per wave; key = (value bits, ~rank) so that larger value, then smaller rank wins
        // (the wave's own winner as ONE reduction of that packed key — what fps_bucket_kernel does — is slower here: 0.455 ->
        // 0.512 us per round at 2048 points, 1.32 -> 1.43 at 16384; max-then-arg keeps each step at one DPP instruction)
        if (lane == 0)
            atomicMax(&best_word[par], ((u64)__float_as_uint(wmax) << 32) | (u64)(0xFFFFFFFFu - wrho));
        if (t == 0) best_word[par ^ 1] = 0ull;
        __syncthreads();
        const unsigned brho = 0xFFFFFFFFu - (unsigned)best_word[par];
        prev_bits = (unsigned)(best_word[par] >> 32);
        prev_rho = brho;
        tmax_prev = tmax;
        if (LDS_XYZ) {
            x1 = lx[brho]; y1 = ly[brho]; z1 = lz[brho];
        } else {
            const int k = fps_rank_to_k(brho, S, bs_shift);
            x1 = dataset[k * 3 + 0]; y1 = dataset[k * 3 + 1]; z1 = dataset[k * 3 + 2];
        }
        if (t == 0) out[r] = (int)brho;
    }
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const unsigned rho = (unsigned)(t + j * THREADS);
        const int k = rho < (unsigned)nslot ? fps_rank_to_k(rho, S, bs_shift) : n;
        if (k < n && tmp) tmp[k] = td[j];
    }
    // ranks -> point indices (kept off the per-round critical path: the division is ~20 dependent instructions)
    __syncthreads();
    for (int r = L + t; r < m; r += THREADS) out[r] = fps_rank_to_k((unsigned)out[r], S, bs_shift);
    if (track) { // smallest round over all lanes (rounds are < 2^31, so the bit patterns order as unsigned)
        note_ties(m - 1); // L < m: at least one round was run
        const unsigned w = ogc_wave_min_u32((unsigned)first_tie);
        __syncthreads();
        if (t == 0) best_word[0] = ~0ull;
        __syncthreads();
        if (lane == 0) atomicMin(&best_word[0], (u64)w);
        __syncthreads();
        if (t == 0) ties_out[b] = (int)(unsigned)best_word[0];
    }
}

__device__ __forceinline__ unsigned fps_rank(int k, int bs_mask, int bs_shift, int S) {
    const unsigned tid = (unsigned)k & (unsigned)bs_mask;
    const unsigned rev = bs_shift ? (__brev(tid) >> (32 - bs_shift)) : 0u;
    return rev * (unsigned)S + ((unsigned)k >> bs_shift);
}


// ---- 4097 .. 8192 points: the same rounds, but only the points a new sample can reach are touched -------------------------
// A round of fps_reg_kernel updates the running minimum of EVERY point (16 points per lane, ~100 VALU instructions per
// wavefront) although, once a few hundred samples are out, a new sample only lowers the minima of the points closer to
// it than their current minimum — a small neighbourhood.  Here the cloud is cut into 64 buckets of 128 points that are
// close in space (a counting sort on a 12-bit interleaved cell key whose bits go to the axes by extent: a kd-tree with
// midpoint splits along the longest axis; any partition would be CORRECT, a compact one prunes well), each owned by one
// wavefront (two points per lane and bucket).  Per bucket the owner keeps the bounding box, the largest running minimum
// and the rank of the point holding it.  A round: (1) lanes 0 .. BPW-1 compute the squared distance from the new sample
// to their bucket's box with the SAME fp32 expression as the point distances — IEEE rounding is monotonic, so it is a
// lower bound of every computed point distance in the box — and a bucket whose bound is not below its maximum keeps all
// its minima (min(d, t) = t for every point); (2) the other buckets are updated and re-reduced; (3) the wavefront's best
// bucket goes to the 64-bit LDS maximum exactly as in fps_reg_kernel.  Values, winners and tie order are those of the
// plain rounds (a skipped update is an update that changes nothing); tests/test_ops_gpu.py runs both against the oracle.
// Clouds with a non-finite coordinate have no box: every bucket is updated every round.
// 1: one reduction of a packed 64-bit key per bucket; 0: max-then-arg with 32-bit DPP steps and one readlane in between (measured,
// 8192 -> 4096 / 8192 -> 2048 x 16: 0.691 / 0.736 us per round against 0.714 / 0.761)
#ifndef FPS_BUCKET_PACKED_KEY
#define FPS_BUCKET_PACKED_KEY 1
#endif
#ifndef FPS_BUCKET_SKIP
// skip a bucket's re-reduction when its maximum is untouched (0: always reduce).  Only with 128 buckets (sixteen per wavefront:
// 16384 -> 4096: 0.975 -> 0.931 us per round); with 64 the wavefront that owns the new sample always re-reduces that sample's
// bucket and sets the round's time, the test only adds to it (8192 -> 2048: 0.757 -> 0.785 us)
#define FPS_BUCKET_SKIP 1
#endif
// NB = 128 buckets (8193 .. 16384 points, sixteen wavefronts): the rank-ordered copy of the cloud no longer fits the LDS next to
// nothing (192 KiB), so the winner's coordinates travel with the reduction instead.  Whenever a bucket is re-reduced the lane
// that owns its new maximum leaves that point's coordinates in the bucket's LDS slot (private to the wavefront); the record lane
// whose bucket is the wavefront's best copies that slot into the wavefront's slot of the ROUND (double-buffered by round parity,
// rewritten by every wavefront every round) next to its 64-bit maximum; the key's low word carries the wavefront's number below
// the rank (ranks are unique, so the passenger bits never decide a comparison), and after the barrier everybody reads the
// winner's slot.  One more LDS round trip in front of the barrier (~50 ns) against an L2 round trip behind it.
template <int NB>
struct FpsBuckets {
    static constexpr int SLOTS = NB * 128;             // points incl. padding
    static constexpr int KEYBITS = NB <= 64 ? 12 : 13; // bins of the counting sort
    static constexpr bool XYZ_LDS = SLOTS <= 8192;     // rank-ordered copy of the cloud in LDS
    static constexpr int WSHIFT = XYZ_LDS ? 0 : 4;     // passenger bits below the rank in a key's low word: the wavefront
};
constexpr int FPSB_SLOTS = FpsBuckets<64>::SLOTS;

template <int WAVES, int NB = 64>
__global__ __launch_bounds__(WAVES *OGC_WAVE) void fps_bucket_kernel(int n, int m, int bs_shift,
                                                                      const float *__restrict__ xyz,
                                                                      float *__restrict__ temp, int *__restrict__ idxs,
                                                                      const int *__restrict__ ties_in,
                                                                      int *__restrict__ ties_out) {
    typedef FpsBuckets<NB> Cfg;
    constexpr int THREADS = WAVES * OGC_WAVE, BPW = NB / WAVES, PTS = 2 * BPW, NBIN = 1 << Cfg::KEYBITS, SLOTS = Cfg::SLOTS;
    constexpr bool XYZ_LDS = Cfg::XYZ_LDS;
    constexpr int WSHIFT = Cfg::WSHIFT;
    static_assert(WAVES <= 16 && (XYZ_LDS || WAVES <= (1 << WSHIFT)), "wavefront number in the key's passenger bits");
    extern __shared__ __attribute__((aligned(16))) float fps_smem[];
    u64 *best_word = reinterpret_cast<u64 *>(fps_smem);           // [2]
    float *red = fps_smem + 4;                                    // [8 * WAVES] reduction scratch
    // !XYZ_LDS: cand[NB] (x, y, z, -) of every bucket's current maximum, wcand[2][WAVES] the wavefronts' best of a round
    float4 *cand = reinterpret_cast<float4 *>(fps_smem + 4 + 8 * 16);
    float4 *wcand = cand + (XYZ_LDS ? 0 : NB);
    float *lx = reinterpret_cast<float *>(wcand + (XYZ_LDS ? 0 : 2 * WAVES)), *ly = lx + SLOTS, *lz = ly + SLOTS; // xyz by RANK slot (XYZ_LDS)
    int *hist = reinterpret_cast<int *>(lx);                      // [NBIN]  (build only: aliases lx)
    unsigned short *perm = reinterpret_cast<unsigned short *>(hist + NBIN); // [SLOTS] sorted position -> point
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
    const float *__restrict__ dataset = xyz + (size_t)b * n * 3;
    float *tmp = temp ? temp + (size_t)b * n : nullptr; // null: see fps_reg_kernel
    int *out = idxs + (size_t)b * m;
    const int S = (n + (1 << bs_shift) - 1) >> bs_shift, bs_mask = (1 << bs_shift) - 1;

    const int L = ties_in ? max(1, min(ties_in[b], m)) : 1; // known prefix of the answer (see fps_reg_kernel)
    if (L >= m) {
        for (int r = t; r < m; r += THREADS) out[r] = r;
        if (t == 0 && ties_out) ties_out[b] = ties_in[b];
        return;
    }
    const bool track = ties_out != nullptr;
    int first_tie = 0x7fffffff;

    // ---- build: bounding box of the cloud
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool finite = true;
    for (int k = t; k < n; k += THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = dataset[k * 3 + a];
            finite = finite && fabsf(v) < INFINITY;
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
    }
    const bool wave_finite = __builtin_amdgcn_ballot_w64(!finite) == 0;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wave * 8 + a] = lo[a]; red[wave * 8 + 3 + a] = hi[a]; }
        red[wave * 8 + 6] = wave_finite ? 1.0f : 0.0f;
    }
    if (t == 0) { best_word[0] = 0ull; best_word[1] = 0ull; }
    __syncthreads();
    bool all_finite = true;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], red[w * 8 + a]); hi[a] = fmaxf(hi[a], red[w * 8 + 3 + a]); }
        all_finite = all_finite && red[w * 8 + 6] != 0.0f;
    }
    const bool always = !all_finite; // no usable boxes: every bucket is updated every round
    // key bits to the axes, most significant first: each bit halves the axis whose cells are longest
    int nbits[3] = {0, 0, 0};
    unsigned seq = 0u; // 2 bits per key bit (most significant key bit in the low bits of seq)
    {
        float cell[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        for (int i = 0; i < Cfg::KEYBITS; ++i) {
            int a = 0;
            if (cell[1] > cell[a]) a = 1;
            if (cell[2] > cell[a]) a = 2;
            seq |= (unsigned)a << (2 * i);
            if (a == 0) { cell[0] *= 0.5f; ++nbits[0]; } else if (a == 1) { cell[1] *= 0.5f; ++nbits[1]; } else { cell[2] *= 0.5f; ++nbits[2]; }
        }
    }
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) inv[a] = (all_finite && hi[a] > lo[a]) ? (float)(1 << nbits[a]) / (hi[a] - lo[a]) : 0.0f;
    for (int e = t; e < NBIN; e += THREADS) hist[e] = 0;
    __syncthreads();
    // ---- counting sort by key: bin counts, then exclusive starts, then positions
    int mykey[PTS], myoff[PTS];
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int k = t + i * THREADS;
        mykey[i] = -1;
        myoff[i] = 0;
        if (k < n) {
            int q[3], rem[3] = {nbits[0], nbits[1], nbits[2]};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = dataset[k * 3 + a];
                int c = (int)((v - lo[a]) * inv[a]);
                q[a] = min(max(c, 0), (1 << nbits[a]) - 1);
            }
            int key = 0;
            for (int j = 0; j < Cfg::KEYBITS; ++j) {
                const int a = (seq >> (2 * j)) & 3;
                int bit;
                if (a == 0) { --rem[0]; bit = (q[0] >> rem[0]) & 1; }
                else if (a == 1) { --rem[1]; bit = (q[1] >> rem[1]) & 1; }
                else { --rem[2]; bit = (q[2] >> rem[2]) & 1; }
                key = (key << 1) | bit;
            }
            mykey[i] = key;
            myoff[i] = atomicAdd(&hist[key], 1);
        }
    }
    __syncthreads();
    {
        constexpr int PER = NBIN / THREADS;
        int c[PER], sum = 0;
#pragma unroll
        for (int i = 0; i < PER; ++i) { c[i] = hist[t * PER + i]; sum += c[i]; }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        int *wsum = reinterpret_cast<int *>(red);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int i = 0; i < PER; ++i) { hist[t * PER + i] = base; base += c[i]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PTS; ++i) {
        const int k = t + i * THREADS;
        if (k < n) perm[hist[mykey[i]] + myoff[i]] = (unsigned short)k;
        else if (k < SLOTS) perm[k] = 0xFFFFu; // sorted positions n .. SLOTS-1 are padding
    }
    __syncthreads();
    // ---- this lane's points: bucket (s, wave) = s * WAVES + wave, sorted positions 128 * bucket + 2 * lane + {0, 1}
    float px[PTS], py[PTS], pz[PTS], td[PTS];
    unsigned rk[PTS];
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const int bucket = (j >> 1) * WAVES + wave;
        const int k = perm[bucket * 128 + 2 * lane + (j & 1)];
        if (k != 0xFFFF) {
            px[j] = dataset[k * 3 + 0];
            py[j] = dataset[k * 3 + 1];
            pz[j] = dataset[k * 3 + 2];
            td[j] = tmp ? tmp[k] : 1e10f;
            rk[j] = (fps_rank(k, bs_mask, bs_shift, S) << WSHIFT) | (unsigned)(XYZ_LDS ? 0 : wave);
        } else {
            px[j] = py[j] = pz[j] = 0.0f;
            td[j] = -1.0f; // padding never wins: real values are >= 0 and min(d, -1) = -1
            rk[j] = 0xFFFFFFFFu;
        }
    }
    __syncthreads(); // everybody has read perm: the region becomes the rank-ordered copy of the cloud
    if constexpr (XYZ_LDS) {
#pragma unroll
        for (int j = 0; j < PTS; ++j) {
            if (rk[j] != 0xFFFFFFFFu) { lx[rk[j]] = px[j]; ly[rk[j]] = py[j]; lz[rk[j]] = pz[j]; }
        }
    }
    if (t == 0) out[0] = 0;
    // known prefix (see fps_reg_kernel): fold samples 0 .. L-2 into the minima of every point, enter at r = L
    for (int s0 = 0; s0 < L - 1; ++s0) {
        const float sx = dataset[s0 * 3], sy = dataset[s0 * 3 + 1], sz = dataset[s0 * 3 + 2];
#pragma unroll
        for (int j = 0; j < PTS; ++j) td[j] = ogc_min_f32(ogc_sqdist(px[j], py[j], pz[j], sx, sy, sz), td[j]);
    }
    for (int r = 1 + t; r < L; r += THREADS) out[r] = r;
    float x1 = dataset[(L - 1) * 3], y1 = dataset[(L - 1) * 3 + 1], z1 = dataset[(L - 1) * 3 + 2];
    // ---- bucket records: lane s < BPW holds the record of the wave's bucket s: box, largest minimum, the rank of the
    // point that has it, and whether a second point of the bucket has it too
    float blo[3] = {0.f, 0.f, 0.f}, bhi[3] = {0.f, 0.f, 0.f}, bmv = -1.0f;
    unsigned bmr = 0xFFFFFFFFu;
    int btie = 0;

    // (records in lanes 48 + s: the wave-wide key maximum below arrives in the last 16-lane row without leaving the vector unit)
    constexpr int REC0 = 48;
    const bool is_rec = lane >= REC0 && lane < REC0 + BPW;
    auto reduce_bucket = [&](int s, float t0, float t1, unsigned r0, unsigned r1, float c0x, float c0y, float c0z, float c1x,
                             float c1y, float c1z) {
#if FPS_BUCKET_PACKED_KEY
        // ONE reduction of the packed key (value bits, ~rank) over the bucket's 128 points instead of max-then-arg: six DPP
        // steps, each two moves, a 64-bit compare and two selects
        const long long k0 = fps_key(t0, r0), k1 = fps_key(t1, r1);
        const long long k = fps_max_to_last_row_i64(k0 > k1 ? k0 : k1);
        const float wm = __int_as_float(__builtin_amdgcn_readlane((int)(k >> 32), 63));
        const float rec_v = __int_as_float((int)(k >> 32));
        const unsigned rec_r = 0xFFFFFFFFu - (unsigned)k;
#else
        // max-then-arg with every step ONE DPP instruction and a single readlane in between (the maximum has to reach all
        // lanes for the comparison); both results arrive in the last row, where the records live
        const int wbits = fps_max_to_last_row_i32(max(__float_as_int(t0), __float_as_int(t1)));
        const float wm = __int_as_float(__builtin_amdgcn_readlane(wbits, 63));
        const unsigned cand = min(t0 == wm ? r0 : 0xFFFFFFFFu, t1 == wm ? r1 : 0xFFFFFFFFu);
        const float rec_v = wm;
        const unsigned rec_r = fps_min_to_last_row_u32(cand);
#endif
        int tie = 0;
        if (track) { // a second point with the winning value?  (only read after the round's barrier)
            const bool h0 = t0 == wm, h1 = t1 == wm;
            tie = __popcll(__builtin_amdgcn_ballot_w64(h0 || h1)) > 1 || __builtin_amdgcn_ballot_w64(h0 && h1) != 0;
        }
        if (lane == REC0 + s) { bmv = rec_v; bmr = rec_r; btie = tie; }
        if constexpr (!XYZ_LDS) { // the owner of the bucket's maximum leaves its coordinates in the bucket's slot
#if FPS_BUCKET_PACKED_KEY
            const unsigned wr = 0xFFFFFFFFu - (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, 63);
#else
            const unsigned wr = (unsigned)__builtin_amdgcn_readlane((int)rec_r, 63);
#endif
            if (r0 == wr) cand[s * WAVES + wave] = make_float4(c0x, c0y, c0z, 0.f);
            else if (r1 == wr) cand[s * WAVES + wave] = make_float4(c1x, c1y, c1z, 0.f);
        }
    };
#pragma unroll
    for (int s = 0; s < BPW; ++s) {
        const bool r0 = rk[2 * s] != 0xFFFFFFFFu, r1 = rk[2 * s + 1] != 0xFFFFFFFFu;
        const float c0[3] = {px[2 * s], py[2 * s], pz[2 * s]}, c1[3] = {px[2 * s + 1], py[2 * s + 1], pz[2 * s + 1]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float mn = fminf(r0 ? c0[a] : INFINITY, r1 ? c1[a] : INFINITY);
            float mx = fmaxf(r0 ? c0[a] : -INFINITY, r1 ? c1[a] : -INFINITY);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                mn = fminf(mn, __shfl_xor(mn, off, 64));
                mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            }
            if (lane == REC0 + s) { blo[a] = mn; bhi[a] = mx; }
        }
        reduce_bucket(s, td[2 * s], td[2 * s + 1], rk[2 * s], rk[2 * s + 1], px[2 * s], py[2 * s], pz[2 * s], px[2 * s + 1],
                      py[2 * s + 1], pz[2 * s + 1]);
    }
    __syncthreads();

    for (int r = L; r < m; ++r) {
        const int par = r & 1;
        // (1) which of the wave's buckets can the new sample reach?  The box distance, rounded like a point distance.
        unsigned reach;
        {
#pragma clang fp contract(off)
            float dx = fmaxf(fmaxf(blo[0] - x1, x1 - bhi[0]), 0.0f);
            float dy = fmaxf(fmaxf(blo[1] - y1, y1 - bhi[1]), 0.0f);
            float dz = fmaxf(fmaxf(blo[2] - z1, z1 - bhi[2]), 0.0f);
            const float d2 = ogc_sqsum3(dx, dy, dz);
            reach = (unsigned)(__builtin_amdgcn_ballot_w64(is_rec && (always || d2 < bmv)) >> REC0);
        }
        // (2) update and re-reduce those
        if (reach != 0u) {
            const ogc_v2f qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
#pragma unroll
            for (int s = 0; s < BPW; ++s) {
                if ((reach >> s) & 1u) { // wave-uniform
                    const ogc_v2f d = fps_sqdist2((ogc_v2f){px[2 * s], px[2 * s + 1]}, (ogc_v2f){py[2 * s], py[2 * s + 1]},
                                                  (ogc_v2f){pz[2 * s], pz[2 * s + 1]}, qx, qy, qz);
                    td[2 * s] = ogc_min_f32(d.x, td[2 * s]);
                    td[2 * s + 1] = ogc_min_f32(d.y, td[2 * s + 1]);
                    // Does the point that holds the bucket's maximum still hold it?  Then the record stands — every other
                    // point only decreases, and none of them had the value — and the re-reduction (six 64-bit DPP steps, most
                    // of a round's dependent chain) is skipped.  (With tie tracking on, only when no second point had the
                    // value: its fate is not looked at here.)  The record comes out of its lane by v_readlane.
                    if constexpr (FPS_BUCKET_SKIP && NB > 64) {
                        const unsigned cur_r = (unsigned)__builtin_amdgcn_readlane((int)bmr, REC0 + s);
                        const unsigned cur_v = (unsigned)__builtin_amdgcn_readlane(__float_as_int(bmv), REC0 + s);
                        const int cur_t = track ? __builtin_amdgcn_readlane(btie, REC0 + s) : 0;
                        const bool holds = (rk[2 * s] == cur_r && __float_as_uint(td[2 * s]) == cur_v) ||
                                           (rk[2 * s + 1] == cur_r && __float_as_uint(td[2 * s + 1]) == cur_v);
                        if (cur_t == 0 && __builtin_amdgcn_ballot_w64(holds) != 0) continue;
                    }
                    reduce_bucket(s, td[2 * s], td[2 * s + 1], rk[2 * s], rk[2 * s + 1], px[2 * s], py[2 * s], pz[2 * s],
                                  px[2 * s + 1], py[2 * s + 1], pz[2 * s + 1]);
                }
            }
        }
        // (3) the wave's best bucket -> LDS maximum over the waves; key = (value bits, ~rank): larger value, then smaller rank.
        // ONE reduction over the packed key of the BPW records (lanes 48 .. 48+BPW-1; the value bits ordered as signed integers like
        // fps_wave_max, empty buckets at -1 never win) instead of max-then-arg: lane 48 ends up holding the winner's key in
        // registers and hands it to the LDS maximum — no readlane / ballot / readlane round trip through the scalar unit.
        static_assert(BPW == 4 || BPW == 8 || BPW == 16, "fps_max_low_lanes_i64 covers 4, 8 or 16 lanes");
        const long long mykey = is_rec ? fps_key(bmv, bmr) : (long long)0x8000000000000000ull;
        const long long key = fps_max_low_lanes_i64<BPW>(mykey);
        if constexpr (XYZ_LDS) {
            if (lane == REC0) atomicMax(&best_word[par], (u64)key);
        } else if (is_rec && mykey == key) { // the record lane of the wavefront's best bucket (s = lane - REC0)
            wcand[par * WAVES + wave] = cand[(lane - REC0) * WAVES + wave];
            atomicMax(&best_word[par], (u64)key);
        }
        if (t == 0) best_word[par ^ 1] = 0ull;
        __syncthreads();
        const u64 best = best_word[par];
        const unsigned brho = 0xFFFFFFFFu - (unsigned)best;
        if constexpr (XYZ_LDS) {
            x1 = lx[brho]; y1 = ly[brho]; z1 = lz[brho];
        } else {
            const float4 c = wcand[par * WAVES + (brho & ((1u << WSHIFT) - 1u))];
            x1 = c.x; y1 = c.y; z1 = c.z;
        }
        if (t == 0) out[r] = (int)(brho >> WSHIFT);
        if (track) { // did a second point attain this round's maximum?  Another bucket with it, or the winner's bucket twice
            const bool eq = is_rec && __float_as_uint(bmv) == (unsigned)(best >> 32);
            const bool tie = eq && (bmr != brho || btie != 0);
            if (__builtin_amdgcn_ballot_w64(tie) != 0 && first_tie > r) first_tie = r;
        }
    }
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        if (rk[j] != 0xFFFFFFFFu && tmp) tmp[fps_rank_to_k(rk[j] >> WSHIFT, S, bs_shift)] = td[j];
    }
    __syncthreads();
    for (int r = L + t; r < m; r += THREADS) out[r] = fps_rank_to_k((unsigned)out[r], S, bs_shift);
    if (track) {
        if (t == 0) best_word[0] = ~0ull;
        __syncthreads();
        if (lane == 0) atomicMin(&best_word[0], (u64)(unsigned)first_tie);
        __syncthreads();
        if (t == 0) ties_out[b] = (int)(unsigned)best_word[0];
    }
}

// Large-N fallback (N > 16384): same rounds, but xyz/temp stay in global memory (L2-resident).
__global__ __launch_bounds__(1024) void fps_mem_kernel(int n, int m, int bs_shift,
                                                       const float *__restrict__ xyz,
                                                       float *__restrict__ temp, int *__restrict__ idxs) {
    __shared__ float s_wmax[2][16];
    __shared__ u64 s_best[2];
    const int t = threadIdx.x;
    const int b = blockIdx.x;
    const float *__restrict__ dataset = xyz + (size_t)b * n * 3;
    float *tmp = temp + (size_t)b * n;
    int *out = idxs + (size_t)b * m;
    const int bs_mask = (1 << bs_shift) - 1;
    const int S = (n + bs_mask) >> bs_shift;
    if (t == 0) {
        out[0] = 0;
        s_best[0] = ~0ull;
        s_best[1] = ~0ull;
    }
    __syncthreads();
    int old = 0;
    for (int r = 1; r < m; ++r) {
        const int par = r & 1;
        const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
        float tmax = -1.0f;
        u64 tbest = ~0ull;
        for (int k = t; k < n; k += 1024) {
            const float d = ogc_sqdist(dataset[k * 3 + 0], dataset[k * 3 + 1], dataset[k * 3 + 2], x1, y1, z1);
            const float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            const u64 key = ((u64)fps_rank(k, bs_mask, bs_shift, S) << 32) | (unsigned)k;
            if (d2 > tmax) { tmax = d2; tbest = key; }
            else if (d2 == tmax && key < tbest) tbest = key;
        }
        const float wmax = ogc_wave_max_f32(tmax);
        if ((t & 63) == 0) s_wmax[par][t >> 6] = wmax;
        __syncthreads();
        float gmax = s_wmax[par][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) gmax = fmaxf(gmax, s_wmax[par][w]);
        if (tmax == gmax) atomicMin(&s_best[par], tbest);
        if (t == 0) s_best[par ^ 1] = ~0ull;
        __syncthreads();
        old = __builtin_amdgcn_readfirstlane((int)(unsigned)s_best[par]);
        if (t == 0) out[r] = old;
    }
}

// Large single clouds (the 10^5-point raw scans of the test / data-preparation pipelines, utils/data_util.py:8-20):
// G workgroups per cloud, each with its slice of the points and their running minimum distances in registers.  A
// round = local update + local winner -> one 64-bit (distance bits, ~rank) word per workgroup in global memory ->
// grid barrier on a per-cloud counter -> every workgroup reduces the G words itself (one lane per word) and fetches
// the winner's coordinates.  The words are double-buffered by round parity; a workgroup cannot run two rounds ahead
// because every round needs everyone's arrival.  Launched cooperatively (all workgroups resident, or the launch
// fails and the one-workgroup kernel above runs instead); the wait is bounded so that a lost workgroup cannot hang
// the device.
constexpr int FPS_COOP_PTS = 4;      // points per thread
constexpr int FPS_COOP_THREADS = 1024;

__global__ __launch_bounds__(FPS_COOP_THREADS) void fps_coop_kernel(int n, int m, int bs_shift, int G,
                                                                    const float *__restrict__ xyz,
                                                                    float *__restrict__ temp, int *__restrict__ idxs,
                                                                    u64 *__restrict__ words /* [b][2][G] */,
                                                                    unsigned *__restrict__ counters /* [b] */) {
    __shared__ u64 s_best[2];
    __shared__ u64 s_winner;
    const int t = threadIdx.x, lane = t & 63;
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const float *__restrict__ dataset = xyz + (size_t)b * n * 3;
    float *tmp = temp + (size_t)b * n;
    int *out = idxs + (size_t)b * m;
    u64 *wb = words + (size_t)b * 2 * G;
    unsigned *counter = counters + b;
    const int bs_mask = (1 << bs_shift) - 1;
    const int S = (n + bs_mask) >> bs_shift;
    const int slice = (n + G - 1) / G;
    const int k0 = g * slice;
    // this thread's points: k0 + t + j * 1024
    float px[FPS_COOP_PTS], py[FPS_COOP_PTS], pz[FPS_COOP_PTS], td[FPS_COOP_PTS];
    unsigned inv_rank[FPS_COOP_PTS];
#pragma unroll
    for (int j = 0; j < FPS_COOP_PTS; ++j) {
        const int k = k0 + t + j * FPS_COOP_THREADS;
        const bool ok = k < min(n, k0 + slice);
        px[j] = ok ? dataset[k * 3] : 0.f;
        py[j] = ok ? dataset[k * 3 + 1] : 0.f;
        pz[j] = ok ? dataset[k * 3 + 2] : 0.f;
        td[j] = ok ? tmp[k] : -1.0f; // padding can never win (distances are >= 0)
        inv_rank[j] = ok ? 0xFFFFFFFFu - fps_rank(k, bs_mask, bs_shift, S) : 0u;
    }
    if (t == 0) { s_best[0] = 0ull; s_best[1] = 0ull; }
    if (g == 0 && t == 0) out[0] = 0;
    __syncthreads();
    float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
    for (int r = 1; r < m; ++r) {
        const int par = r & 1;
        u64 best = 0ull;
#pragma unroll
        for (int j = 0; j < FPS_COOP_PTS; ++j) {
            const float d = ogc_sqdist(px[j], py[j], pz[j], x1, y1, z1);
            if (td[j] >= 0.0f) td[j] = ogc_min_f32(d, td[j]);
            const u64 key = ((u64)__float_as_uint(fmaxf(td[j], 0.0f)) << 32) | inv_rank[j];
            if (td[j] >= 0.0f && key > best) best = key;
        }
        // wave max of the 64-bit key, then one LDS atomic per wave
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)best, off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
            const u64 o = ((u64)hi << 32) | lo;
            if (o > best) best = o;
        }
        if (lane == 0) atomicMax(&s_best[par], best);
        if (t == 0) s_best[par ^ 1] = 0ull;
        __syncthreads();
        if (t < 64) { // first wave: publish, barrier, reduce the G words
            if (t == 0) {
                __hip_atomic_store(&wb[par * G + g], s_best[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = (unsigned)G * (unsigned)r;
                long long spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1ll << 22))
                    ++spins;
            }
            __builtin_amdgcn_wave_barrier();
            u64 wv = 0ull;
            for (int i = lane; i < G; i += 64) {
                const u64 v = __hip_atomic_load(&wb[par * G + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v > wv) wv = v;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)wv, off, 64), hi = __shfl_xor((unsigned)(wv >> 32), off, 64);
                const u64 o = ((u64)hi << 32) | lo;
                if (o > wv) wv = o;
            }
            if (t == 0) s_winner = wv;
        }
        __syncthreads();
        const unsigned rho = 0xFFFFFFFFu - (unsigned)s_winner;
        const int k = fps_rank_to_k(rho, S, bs_shift);
        x1 = dataset[k * 3]; y1 = dataset[k * 3 + 1]; z1 = dataset[k * 3 + 2];
        if (g == 0 && t == 0) out[r] = k;
    }
#pragma unroll
    for (int j = 0; j < FPS_COOP_PTS; ++j) {
        const int k = k0 + t + j * FPS_COOP_THREADS;
        if (k < min(n, k0 + slice)) tmp[k] = td[j];
    }
}

// cuda_utils.h:10-14 opt_n_threads(), evaluated through double log() exactly like the reference.
int fps_ref_block_shift(int work_size) {
    int pow_2 = (int)(log((double)work_size) / log(2.0));
    if (pow_2 > 10) pow_2 = 10;
    if (pow_2 < 0) pow_2 = 0;
    return pow_2;
}

// Raises a kernel's dynamic-LDS cap (needed above 64 KiB), once per process, DEVICE and kernel (the attribute belongs to the
// kernel's code object on one device: a process that drives several GPUs must raise it on each); false when the runtime
// refuses — the caller then launches a variant that needs less.
static bool fps_allow_lds(const void *kernel, size_t lds) {
    if (lds <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    struct Entry { const void *kernel; int dev; size_t lds; bool ok; };
    static Entry table[128];
    static int used = 0;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used; ++i)
        if (table[i].kernel == kernel && table[i].dev == dev && table[i].lds >= lds) return table[i].ok;
    const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) (void)hipGetLastError(); // (the refusal is handled: it must not be reported by the next launch check)
    if (used < 128) table[used++] = Entry{kernel, dev, lds, ok};
    return ok;
}

template <int PTS, int THREADS>
void fps_launch(int b, int n, int m, int shift, const float *xyz, float *temp, int *idx, const int *ties_in,
                int *ties_out, hipStream_t stream) {
    constexpr int slots = PTS * THREADS;
    constexpr bool lds_xyz = slots <= FPS_LDS_XYZ_MAX;
    constexpr size_t lds = 8 * sizeof(float) + (lds_xyz ? 3 * slots * sizeof(float) : 0);
    auto kern = fps_reg_kernel<PTS, THREADS, lds_xyz>;
    if (!fps_allow_lds(reinterpret_cast<const void *>(kern), lds)) {
        // no room for the rank-ordered copy of the cloud in LDS: the same rounds reading the winner's coordinates from memory
        hipLaunchKernelGGL((fps_reg_kernel<PTS, THREADS, false>), dim3(b), dim3(THREADS), 8 * sizeof(float), stream, n, m, shift, xyz,
                           temp, idx, ties_in, ties_out);
        return;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(THREADS), lds, stream, n, m, shift, xyz, temp, idx, ties_in, ties_out);
}

} // namespace

static int fps_impl(const char *name, int b, int n, int m, const float *xyz, float *temp, int *idx, const int *ties_in,
                    int *ties_out, ogc_stream_t stream, bool allow_null_temp) {
    OGC_REQUIRE(b >= 0 && n >= 0 && m >= 0, "%s: negative dimension", name);
    if (b == 0 || m <= 0) return OGC_OK; // sampling_gpu.cu:98
    OGC_REQUIRE(n >= 1, "%s: n must be >= 1 when m > 0", name);
    OGC_REQUIRE(xyz && idx, "%s: null pointer", name);
    OGC_REQUIRE((long long)b * n * 3 < (1ll << 31), "%s: xyz exceeds 32-bit indexing", name);
    const int shift = fps_ref_block_shift(n);
    const int bs = 1 << shift;
    const int slots = ((n + bs - 1) / bs) * bs; // rank slots = bs * ceil(n / bs)
    // temp == null (the chain entry point only): the running minima live in registers for the whole run, start at the reference's
    // 1e10 (pointnet2.py:33) and are not written back — no fill launch in front of the sampling, no (b, n) buffer
    OGC_REQUIRE(temp || (allow_null_temp && slots <= 16384), "%s: temp is required%s", name,
                allow_null_temp ? " above 16384 points (the minima are kept in memory there)" : "");
    hipStream_t s = (hipStream_t)stream;
    if (slots <= 64) fps_launch<1, 64>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 128) fps_launch<2, 64>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 256) fps_launch<4, 64>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 512) fps_launch<2, 256>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 1024) fps_launch<4, 256>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 2048) fps_launch<8, 256>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 4096) fps_launch<8, 512>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    else if (slots <= 8192) {
        // bucketed rounds (fps_bucket_kernel) pay from a few hundred samples on; OGC_FPS_BUCKETS=0: the plain rounds
        // sixteen wavefronts (four buckets each) where the rounds are the caller's critical path and the chip is otherwise idle
        // (a pair of clouds: FlowStep3D at B = 1: 0.711 -> 0.678 us per round); eight for batches, whose launches ride a side
        // stream underneath dense kernels and must find room next to them
        static const char *bk = getenv("OGC_FPS_BUCKETS");
        const int mode = bk ? atoi(bk) : (b <= 4 ? 16 : 8);
        const size_t lds = (4 + 128 + 3 * FPSB_SLOTS) * sizeof(float);
        const void *bucket_kernel = mode == 16 ? reinterpret_cast<const void *>(fps_bucket_kernel<16>)
                                  : mode == 8  ? reinterpret_cast<const void *>(fps_bucket_kernel<8>)
                                               : reinterpret_cast<const void *>(fps_bucket_kernel<4>);
        if (mode > 0 && m >= 256 && fps_allow_lds(bucket_kernel, lds)) {
            if (mode == 16) {
                hipLaunchKernelGGL(fps_bucket_kernel<16>, dim3(b), dim3(1024), lds, s, n, m, shift, xyz, temp, idx, ties_in, ties_out);
            } else if (mode == 8) {
                hipLaunchKernelGGL(fps_bucket_kernel<8>, dim3(b), dim3(512), lds, s, n, m, shift, xyz, temp, idx, ties_in, ties_out);
            } else {
                hipLaunchKernelGGL(fps_bucket_kernel<4>, dim3(b), dim3(256), lds, s, n, m, shift, xyz, temp, idx, ties_in, ties_out);
            }
        } else
            fps_launch<16, 512>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    }
    else if (slots <= 16384) {
        // bucketed rounds with 128 buckets on eight wavefronts (OGC_FPS_BUCKETS=16: sixteen; 0: the plain rounds)
        static const char *bk = getenv("OGC_FPS_BUCKETS");
        const int mode = bk ? atoi(bk) : 8;
        typedef FpsBuckets<128> Cfg;
        const size_t lds = (4 + 128) * sizeof(float) + (128 + 2 * 16) * sizeof(float4) +
                           (1 << Cfg::KEYBITS) * sizeof(int) + Cfg::SLOTS * sizeof(unsigned short);
        const void *bucket_kernel = mode == 16 ? reinterpret_cast<const void *>(fps_bucket_kernel<16, 128>)
                                               : reinterpret_cast<const void *>(fps_bucket_kernel<8, 128>);
        // (the runtime refusing the > 64 KiB LDS these need: the plain rounds, fps_launch<32, 512>)
        if (mode > 0 && m >= 256 && fps_allow_lds(bucket_kernel, lds)) {
            if (mode == 16) {
                hipLaunchKernelGGL((fps_bucket_kernel<16, 128>), dim3(b), dim3(1024), lds, s, n, m, shift, xyz, temp, idx, ties_in, ties_out);
            } else {
                hipLaunchKernelGGL((fps_bucket_kernel<8, 128>), dim3(b), dim3(512), lds, s, n, m, shift, xyz, temp, idx, ties_in, ties_out);
            }
        } else
            fps_launch<32, 512>(b, n, m, shift, xyz, temp, idx, ties_in, ties_out, s);
    }
    else {
        // (these kernels do not track ties: a chain through them never takes the shortcut afterwards)
        if (ties_out && ogc_zero_async(ties_out, sizeof(int) * (size_t)b, s) != hipSuccess) {
            ogc_set_error("%s: memset failed", name);
            return OGC_ERR_LAUNCH;
        }
        // cooperative multi-workgroup kernel when the whole job fits on the chip at once, else one workgroup per cloud
        static const char *mode = getenv("OGC_FPS_LARGE"); // "single": development override
        const int G = ogc_divup(n, FPS_COOP_PTS * FPS_COOP_THREADS);
        bool done = false;
        if (!(mode && mode[0] == 's') && (long long)b * G <= 256 && G <= 64 * 16) {
            const size_t bytes_words = sizeof(u64) * (size_t)b * 2 * G;
            char *ws = static_cast<char *>(ogc_workspace(s, bytes_words + sizeof(unsigned) * b));
            if (ws && ogc_zero_async(ws, bytes_words + sizeof(unsigned) * b, s) == hipSuccess) {
                u64 *words = reinterpret_cast<u64 *>(ws);
                unsigned *counters = reinterpret_cast<unsigned *>(ws + bytes_words);
                int gg = G;
                void *args[] = {&n, &m, const_cast<int *>(&shift), &gg, &xyz, &temp, &idx, &words, &counters};
                const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void *>(fps_coop_kernel),
                                                                dim3(b * G), dim3(FPS_COOP_THREADS), args, 0, s);
                if (e == hipSuccess) done = true;
                else (void)hipGetLastError();
            }
        }
        if (!done) hipLaunchKernelGGL(fps_mem_kernel, dim3(b), dim3(1024), 0, s, n, m, shift, xyz, temp, idx);
    }
    OGC_CHECK_LAUNCH(name);
    return OGC_OK;
}

extern "C" int ogc_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                           ogc_stream_t stream) {
    return fps_impl("ogc_furthest_point_sampling", b, n, m, xyz, temp, idx, nullptr, nullptr, stream, false);
}

extern "C" int ogc_furthest_point_sampling_chain(int b, int n, int m, const float *xyz, float *temp, int *idx,
                                                 const int *ties_in, int *ties_out, ogc_stream_t stream) {
    return fps_impl("ogc_furthest_point_sampling_chain", b, n, m, xyz, temp, idx, ties_in, ties_out, stream, true);
}
