// Particle-mixture log-prob kernels for gfx950 (MI355X): forward, backward and the fused
// PPO forward+loss+backward, i.e. HIP units K1 / K2 / K7 of DESIGN.md.
//
// Replaces the TF op subgraph of MixtureGaussianDistribution.log_prob
// (reference networks/utils.py:108-134, op order in SURVEY.md Appendix A) and its autodiff,
// plus the clipped-PPO loss head (reference networks/actor_critic/ppo.py:31-54,
// actor_critic.py:151-156).
//
// Three data paths (chosen by shape in the C entry points at the end of this file):
//  * "tile" (N <= 64, A <= 44, A*N % 4 == 0; templated on N: the shipped N=35, the default N=50, 16, 8):
//    an LDS ring -- LDS-DMA loader waves stream 4-row tiles into LDS planes, compute waves own four action
//    dims of the tile's four rows (4 lanes per segment) and run forward and backward out of LDS and
//    registers, a row wave runs the row stage and the loss head, storer waves copy finished planes back.
//    Because a block owns complete rows, the row-level log-prob, the PPO ratio/clip and the backward run in
//    ONE pass: logits are read from HBM once and dlogits written once (2L traffic instead of 3L).  See the
//    block comment above k_mix_tile.
//  * "wide" (N >= 64, N % 4 == 0: the 128 x 256 stress shape): one wavefront per segment, float4 per
//    lane straight from HBM, 4 rows reduced together by a transposing DPP butterfly.
//  * "wave" (any shape): one 64-lane wavefront owns one segment, lanes stride the particles; a
//    wave keeps a fixed action dim so dloc/dlogstd accumulate in registers.
// Reductions over the batch (dloc, dlogstd, loss sums) go through per-block partial slabs and
// a fixed-order finalize kernel: results are bitwise reproducible (no float atomics).
#include "pfpn_common.h"
#include <stdlib.h>
#include <atomic>

namespace pfpn {

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_PPO = 2 };

typedef float v2f __attribute__((ext_vector_type(2)));     // packed-fp32 operand (v_pk_*_f32)
typedef float f4v __attribute__((ext_vector_type(4)));

struct PpoArgs {
    const float* logp_old;   // [B]
    const float* adv;        // [B]
    const float* value;      // [B]
    const float* value_old;  // [B]
    const float* adv_stats;  // [4] mean, 1/(std+1e-8), clip lo, clip hi   (device)
    float inv_B;             // 1/B (rank-local batch, reference per-worker minibatch mean)
    float clip_lo, clip_hi;  // 1-eps, 1+eps
    float value_coef;        // 0.5
    float* dvalue;           // [B] out
    float* ratio_out;        // [B] out (optional, may be null)
    int want_db;             // also emit column sums of dlogits (fc_policy bias gradient) into the slab
};

constexpr int TK_G = 4;                              // lanes per (row, action-dim) segment
constexpr size_t TK_MAX_LDS = 150 * 1024;            // dynamic LDS the tile kernel may ask for (160 KB per CU)

#ifndef TK_DEPTH
#define TK_DEPTH 2                                   // tiles a loader keeps in flight
#endif

// LDS-DMA of one wave-instruction: lane i's 16 (or 4) bytes at `gsrc` land at lds_byte + 16*i (4*i).
// M0 carries the wave-uniform LDS base and is compiler-reserved: saved/restored inside the statement.
// hipcc does not see these loads: their completion is counted by hand (wait_vmcnt below).
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}
// s_waitcnt vmcnt(n) for a wave-uniform run-time n in [0, 63] (the immediate must be a literal)
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 22: asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); break;
        case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 25: asm volatile("s_waitcnt vmcnt(25)" ::: "memory"); break;
        case 26: asm volatile("s_waitcnt vmcnt(26)" ::: "memory"); break;
        case 27: asm volatile("s_waitcnt vmcnt(27)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        case 29: asm volatile("s_waitcnt vmcnt(29)" ::: "memory"); break;
        case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
        case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
        case 32: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
        case 33: asm volatile("s_waitcnt vmcnt(33)" ::: "memory"); break;
        case 34: asm volatile("s_waitcnt vmcnt(34)" ::: "memory"); break;
        case 35: asm volatile("s_waitcnt vmcnt(35)" ::: "memory"); break;
        case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
        case 37: asm volatile("s_waitcnt vmcnt(37)" ::: "memory"); break;
        case 38: asm volatile("s_waitcnt vmcnt(38)" ::: "memory"); break;
        case 39: asm volatile("s_waitcnt vmcnt(39)" ::: "memory"); break;
        case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
        case 41: asm volatile("s_waitcnt vmcnt(41)" ::: "memory"); break;
        case 42: asm volatile("s_waitcnt vmcnt(42)" ::: "memory"); break;
        case 43: asm volatile("s_waitcnt vmcnt(43)" ::: "memory"); break;
        case 44: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
        case 45: asm volatile("s_waitcnt vmcnt(45)" ::: "memory"); break;
        case 46: asm volatile("s_waitcnt vmcnt(46)" ::: "memory"); break;
        case 47: asm volatile("s_waitcnt vmcnt(47)" ::: "memory"); break;
        case 48: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        case 49: asm volatile("s_waitcnt vmcnt(49)" ::: "memory"); break;
        case 50: asm volatile("s_waitcnt vmcnt(50)" ::: "memory"); break;
        case 51: asm volatile("s_waitcnt vmcnt(51)" ::: "memory"); break;
        case 52: asm volatile("s_waitcnt vmcnt(52)" ::: "memory"); break;
        case 53: asm volatile("s_waitcnt vmcnt(53)" ::: "memory"); break;
        case 54: asm volatile("s_waitcnt vmcnt(54)" ::: "memory"); break;
        case 55: asm volatile("s_waitcnt vmcnt(55)" ::: "memory"); break;
        case 56: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
        case 57: asm volatile("s_waitcnt vmcnt(57)" ::: "memory"); break;
        case 58: asm volatile("s_waitcnt vmcnt(58)" ::: "memory"); break;
        case 59: asm volatile("s_waitcnt vmcnt(59)" ::: "memory"); break;
        case 60: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
        case 61: asm volatile("s_waitcnt vmcnt(61)" ::: "memory"); break;
        case 62: asm volatile("s_waitcnt vmcnt(62)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
    }
}

__device__ __forceinline__ int lds_counter(const int* c) {
    return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Diagnostic phase stamps (s_memtime) of block 0: enabled only when pfpn_debug_stamps() installed a buffer.
// Layout: [wave][slot], slot 0 = count; never read by the kernel itself.
__device__ unsigned long long* g_stamps = nullptr;
#ifdef PFPN_ENABLE_STAMPS
#define PFPN_STAMP(id_)                                                                              \
    do {                                                                                             \
        if (stamps && blockIdx.x == 0 && lane == 0) stamps[wave * 64 + (id_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PFPN_STAMP(id_) do { } while (0)      /* build with -DPFPN_ENABLE_STAMPS for tools/stamp_probe.py */
#endif

// ---------------------------------------------------------------------------------------
// tile path: loader / compute / row / storer waves around an LDS ring
//
// A tile is R = 4 whole batch rows, laid out in an LDS plane exactly as in HBM; a block walks a contiguous run
// of tiles through a ring of P planes.  One block per CU, one start-up barrier, no other block-wide wait.
//
// * LOADER waves (no arithmetic) move tiles HBM -> LDS with LDS-DMA (`global_load_lds_dwordx4`: 1 KiB per
//   wave-instruction, no VGPR destination), together with the tile's action values and per-row scalars.  A loader's
//   instruction stream is nothing but DMAs, so its `s_waitcnt vmcnt(n)` is counted exactly: it keeps two tiles in
//   flight and bumps the plane's `ready` counter when the OLDEST has landed.  The bytes in flight per CU are not
//   limited by the VGPRs of the computing waves (one tile ahead in registers = 20 KB per CU measured ~2.6 TB/s
//   chip-wide).
// * COMPUTE wave w owns action dims [4w, 4w+4) of ALL four rows of every tile: lane = 16 d + 4 r + g (d dim, r row,
//   g particle quarter; particles g, g+4, ...).  It transforms its part of the plane in place (logits -> softmax
//   numerators -> dlogits): no global load, hence no vmcnt wait, anywhere in its loop.  The 4 lanes of a segment
//   reduce with two quad-permute DPP steps.  The particle bank of a lane's (a, n) set is tile-invariant and lives
//   in registers.  The four lanes that hold the same (a, n) set for the four rows sit in one 16-lane DPP row, 4 lanes
//   apart, so the batch sums (dloc, dlogstd, fc_policy bias gradient) are finished after the last tile with two
//   row_ror DPP adds per value and leave straight from registers as the block's slab (fixed order: bitwise
//   reproducible, no float atomics, no LDS staging, no barrier).
// * The ROW wave owns the only cross-wave dependency, the row log-prob: compute waves publish per-segment log p
//   and bump a per-row arrival counter; the row wave sums a row's A values in a fixed order, runs the loss head and
//   publishes {tile number, d loss / d logp} as ONE 8-byte LDS word per row, which the compute waves poll: the
//   gradient arrives with the flag.  The loop is software-pipelined -- forward(k+1) | wait row(k) | backward(k) -- so
//   the row stage of tile k overlaps the forward of tile k+1.
// * STORER waves (modes that write dlogits) copy a plane whose backward is complete LDS -> HBM: whole 1-KiB wave
//   stores, line-aligned whatever the wave count (wave-sized slices of 2,240 B split 128-B lines: 1.07x traffic),
//   non-temporal, then hand the plane back to the loaders.
// * START-UP: nothing of the first tile goes through the loaders -- see the comment in the kernel.
//
// What the phase stamps of block 0 (tools/stamp_probe.py) say at 4096 x 36 x 35 (4 tiles per block): ~1,500 cycles
// until the kernel arguments are there, tile 0 landed at ~4,900 (one cold 5 MB burst from 256 CUs), ~3,600 cycles per
// tile -- set by the third compute wave of SIMD 0 (nine compute waves on four SIMDs), whose forward takes 2,000-2,200
// cycles against 1,400 for a SIMD's first wave -- and the last plane stored ~2,000 cycles after the last backward.
// ---------------------------------------------------------------------------------------
constexpr int TK_R = 4;                              // rows per tile
constexpr int TK_NL = 2;                             // loader waves
constexpr int TK_NS = 2;                             // storer waves (modes that write dlogits)

template <int N, int MODE>
__global__ __launch_bounds__(1024) void k_mix_tile(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,      // MODE_BWD: [B]
    int B, int A, int tanh_mode, int P,      // P: LDS planes in the ring
    int tbase, int trem,                     // tiles per block = tbase (+1 for the first trem blocks)
    float* __restrict__ logp_out,            // [B]   (FWD, PPO)
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // [B,A,N] (BWD, PPO)
    float* __restrict__ dx_out,              // [B,A] optional (BWD)
    float* __restrict__ partial,             // [grid][nq*A*N + 4] (BWD, PPO)
    PpoArgs ppo)
{
    constexpr int R = TK_R;
    constexpr int NP = (N + TK_G - 1) / TK_G;             // particles per lane (9 for N=35)
    constexpr int NPK = (NP - 1) / 2;                     // particle pairs that always exist (packed fp32 math)
    constexpr int NRW = (MODE != MODE_BWD) ? 1 : 0;       // row wave
    constexpr int NS = (MODE != MODE_FWD) ? TK_NS : 0;    // storer waves
    extern __shared__ float4 smem4[];
    const int AN = A * N;
    const int nseg = R * A;                               // segments per full tile
    const int NC = (A + 3) >> 2;                          // compute waves
    const int TS = R * AN;                                // floats per plane
    const int XS = (nseg + 63) & ~63;                     // floats per action-value plane
    // LDS map (floats):  [P][TS] tiles | [P][XS] action values | [P][64] row scalars | [4][nseg] segment log p
    //                    | [4][R] row results (64-bit: tile number + 1 | d loss / d logp) | ints: [4][R] row counters
    //                    | [P] ready | [P] freed | [P] bdone | [3][AN] particle bank
    float* tile = reinterpret_cast<float*>(smem4);
    float* xpl = tile + (size_t)P * TS;
    float* rpl = xpl + P * XS;                            // [P][4][16]: adv, logp_old, value, value_old (PPO) | dlogp (BWD)
    float* segbuf = rpl + P * 64;
    // row results: ONE 8-byte LDS word per row and tile (set k & 3) carries {k + 1, d loss / d logp}: a compute wave
    // polls the word of its row and has the gradient the moment it sees the tile number -- no separate "done" counter
    // behind a second LDS round trip
    unsigned long long* gslot = reinterpret_cast<unsigned long long*>(segbuf + 4 * nseg);     // (8-byte aligned: all terms even)
    int* cnt = reinterpret_cast<int*>(gslot + 4 * R);     // [4][R]: segment log-probs of row r of tile k that have arrived (set k & 3)
    int* ready = cnt + 4 * R;
    int* freed = ready + P;
    int* bdone = freed + P;                               // compute waves done with the backward of the tile in this plane
    float* bank = reinterpret_cast<float*>(bdone + P);    // [3][AN]: loc | isc2 | 64 - ct2

    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
#ifdef PFPN_ENABLE_STAMPS
    unsigned long long* stamps = g_stamps;
#endif
    PFPN_STAMP(0);
    const int T = tbase + ((int)blockIdx.x < trem ? 1 : 0);                    // my tiles: a contiguous run
    const int tile0 = (int)blockIdx.x * tbase + min((int)blockIdx.x, trem);

    // per-tile DMA items: tile chunks of 64 float4 | action-value chunks of 64 floats | one row-scalar piece
    const int nct = (R * AN / 4 + 63) >> 6;
    const int ncx = (nseg + 63) >> 6;
    const int nrs = (MODE == MODE_FWD) ? 0 : 1;
    const int M = nct + ncx + nrs;
    const unsigned tile_b = (unsigned)(uintptr_t)tile, xpl_b = (unsigned)(uintptr_t)xpl, rpl_b = (unsigned)(uintptr_t)rpl;
    // one item of tile `kt` (block-local number) into plane `slot`; short tiles clamp their addresses, never skip
    auto issue_item = [&](int kt, int it_, int slot) __attribute__((always_inline)) {
        const int row0 = (tile0 + kt) * R, rows = min(R, B - row0);
        if (it_ < nct) {
            const int nv4 = rows * AN / 4;
            const int i4 = it_ * 64 + lane;
            if (i4 < R * AN / 4)                               // never past the plane
                dma16(reinterpret_cast<const float4*>(logits + (size_t)row0 * AN) + min(i4, nv4 - 1),
                      __builtin_amdgcn_readfirstlane(tile_b + (unsigned)(slot * TS * 4 + it_ * 1024)));
        } else if (it_ < nct + ncx) {
            const int c = it_ - nct;
            const int i = c * 64 + lane;
            dma4(x + (size_t)row0 * A + min(i, rows * A - 1),
                 __builtin_amdgcn_readfirstlane(xpl_b + (unsigned)((slot * XS + c * 64) * 4)));
        } else {
            // lane j -> array j/16, row j%16
            const int arr = lane >> 4, rw = min(lane & 15, rows - 1);
            const float* src;
            if (MODE == MODE_PPO)
                src = (arr == 0 ? ppo.adv : arr == 1 ? ppo.logp_old : arr == 2 ? ppo.value : ppo.value_old) + row0 + rw;
            else
                src = dlogp_in + row0 + rw;
            dma4(src, __builtin_amdgcn_readfirstlane(rpl_b + (unsigned)(slot * 64 * 4)));
        }
    };

    // Start-up.  Nothing of the first tile goes through the loaders: compute wave w fetches what IT needs -- the
    // particle tables of its four dims (<= 4 N consecutive elements, <= BKU per lane), its action values, and its own
    // four row pieces of tile 0 by LDS-DMA -- in its first instructions, so the first forward pass waits for one HBM
    // round trip and for no other wave.  The one block barrier only publishes the zeroed counters and comes early.
    constexpr int BKU = (4 * N + 63) / 64;                                    // table elements per lane
    float bank_ls[BKU], bank_lc[BKU];
    float x0 = 0.f, g0 = 0.f;                                                 // tile 0: my action value, my row's dlogp (BWD)
    if (wave < NC) {
        const int ndw_ = min(4, A - wave * 4);
        const int e0 = wave * 4 * N, ne = ndw_ * N;
#pragma unroll
        for (int u = 0; u < BKU; ++u) {
            const int i = e0 + min(lane + 64 * u, ne - 1);
            bank_ls[u] = logstd[i];
            bank_lc[u] = loc[i];
        }
        if (T > 0) {
            const int row0 = tile0 * R, rows0 = min(R, B - row0);
            const int r_ = (lane >> 2) & 3, a_ = min(wave * 4 + (lane >> 4), A - 1);
            if (r_ < rows0) {
                x0 = x[(size_t)(row0 + r_) * A + a_];
                if (MODE == MODE_BWD) g0 = dlogp_in[row0 + r_];
            }
            // my piece of row rr: ndw N floats = ndw N / 4 float4 (AN % 4 == 0 makes every piece whole), one DMA each
            const int nf4 = ne >> 2;
            for (int rr = 0; rr < rows0; ++rr)
                if (lane < nf4)
                    dma16(reinterpret_cast<const float4*>(logits + (size_t)(row0 + rr) * AN + e0) + lane,
                          __builtin_amdgcn_readfirstlane(tile_b + (unsigned)((rr * AN + e0) * 4)));
        }
    } else if (NRW && wave == NC + TK_NL) {
        if (MODE == MODE_PPO && T > 0) issue_item(0, nct + ncx, 0);           // row wave: tile 0's row scalars
    } else if (wave == NC) {
        // first loader: zero gslot (8 R ints), cnt, ready, freed, bdone; ready[0] = landed (tile 0 is the compute waves')
        for (int i = lane; i < 12 * R + 3 * P; i += 64)
            reinterpret_cast<int*>(gslot)[i] = (i == 12 * R && T > 0) ? TK_NL : 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PFPN_STAMP(42);

    // =========================== loader waves ================================================
    if (wave >= NC && wave < NC + TK_NL) {
        const int l = wave - NC;
        const int Q = (M - l + TK_NL - 1) / TK_NL;         // my DMAs per tile from tile 1 on (constant)
        int dmax = TK_DEPTH;                               // tiles a loader keeps in flight
        while (dmax > 1 && (dmax - 1) * Q > 63) --dmax;
        if (dmax > P) dmax = P;
        const int nfree = (MODE == MODE_FWD) ? NC : NS;    // who hands a plane back
        int issued = (T > 0) ? 1 : 0, signalled = issued;  // tile 0 is fetched by the compute waves themselves
        int slot_i = (P > 1) ? 1 : 0, wrap_i = (P > 1) ? 0 : 1, slot_s = slot_i;
        if (T == 0) { slot_i = 0; wrap_i = 0; slot_s = 0; }
        auto issue_tile = [&]() __attribute__((always_inline)) {
            const int row0 = (tile0 + issued) * R, rows = min(R, B - row0);
            const int slot = slot_i;
            if (rows == R) {
                // full tile: per-DMA overhead is a pointer bump (a wave issues at most one instruction per ~4 cycles)
                const char* gp = reinterpret_cast<const char*>(logits + (size_t)row0 * AN) + (size_t)(l * 64 + lane) * 16;
                unsigned lb = __builtin_amdgcn_readfirstlane(tile_b + (unsigned)(slot * TS * 4 + l * 1024));
                int c = l;
                for (; c < nct - 1; c += TK_NL) {
                    dma16(gp, lb);
                    gp += TK_NL * 1024;
                    lb += TK_NL * 1024;
                }
                if (c == nct - 1 && c * 64 + lane < R * AN / 4) dma16(gp, lb);       // last chunk: partial (never empty)
                for (int it_ = nct + ((l - nct) % TK_NL + TK_NL) % TK_NL; it_ < M; it_ += TK_NL) issue_item(issued, it_, slot);
            } else {
                for (int it_ = l; it_ < M; it_ += TK_NL) issue_item(issued, it_, slot);
            }
            ++issued;
            if (++slot_i == P) { slot_i = 0; ++wrap_i; }
        };
        __builtin_amdgcn_s_setprio(3);                     // (loaders share SIMDs with compute waves)
        while (signalled < T) {
            bool can = issued < T && issued - signalled < dmax;
            if (can && wrap_i > 0)                         // plane still in use?
                can = lds_counter(&freed[slot_i]) >= nfree * wrap_i;
            if (can) {
                issue_tile();
            } else if (issued > signalled) {
                wait_vmcnt(Q * (issued - signalled - 1));  // oldest unsignalled tile: everything after it may stay in flight
                if (lane == 0) atomicAdd(&ready[slot_s], 1);
                ++signalled;
                if (++slot_s == P) slot_s = 0;
                if (signalled <= 4) PFPN_STAMP(4 + (signalled - 1) * 6);
            } else {
                __builtin_amdgcn_s_sleep(2);               // ring full and nothing of mine in flight
            }
        }
        return;
    }

    // =========================== row wave ====================================================
    if (NRW && wave == NC + TK_NL) {
        float pol_sum = 0.f, val_sum = 0.f;
        float adv_mean = 0.f, adv_rstd = 1.f, adv_lo = -INFINITY, adv_hi = INFINITY;
        if (MODE == MODE_PPO) {                            // (loads in flight across the start-up wait)
            adv_mean = ppo.adv_stats[0]; adv_rstd = ppo.adv_stats[1];
            adv_lo = ppo.adv_stats[2]; adv_hi = ppo.adv_stats[3];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile 0's row scalars (my own DMA)
        __builtin_amdgcn_s_setprio(3);     // every compute wave waits for this wave's few instructions once per tile
        const int kq = lane >> 4, jl = lane & 15;
        int slot = 0, lap = 0;
        for (int k = 0; k < T; ++k) {
            const int row0 = (tile0 + k) * R;
            const int rows = min(R, B - row0);
            const float* sb = segbuf + (k & 3) * nseg;
            const int rk = kq;                             // 16-lane group kq owns row kq
            const bool rk_ok = rk < rows;
            const int b = row0 + rk;
            const bool owner = rk_ok && jl == 0;
            // the row's scalars arrive with the tile: everything that does not need the log-prob happens before the
            // wait for the forward passes
            float adv = 0.f, lpo = 0.f, nadv = 0.f, dvv = 0.f;
            if (MODE == MODE_PPO) {
                while (lds_counter(&ready[slot]) < TK_NL * (lap + 1)) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                const float* rs = rpl + slot * 64;
                adv = rs[rk]; lpo = rs[16 + rk];
                const float val_b = rs[32 + rk], vold_b = rs[48 + rk];
                nadv = clip_adv((adv - adv_mean) * adv_rstd, adv_lo, adv_hi);
                dvv = val_b - (adv + vold_b);
                if (owner) {
                    val_sum += dvv * dvv;
                    ppo.dvalue[b] = 2.f * dvv * ppo.value_coef * ppo.inv_B;
                }
            }
            // every row of the tile has all its A segment log-probs (lane j waits for row j)
            if (lane < rows) {
                const int target = A * ((k >> 2) + 1);
                while (lds_counter(&cnt[(k & 3) * R + lane]) < target) { }
            }
            asm volatile("" ::: "memory");
            // the group sums its row's A segment log-probs (fixed lanes, fixed order: reproducible); the rotation steps
            // associate differently per lane, so only the group's lane 0 -- the owner -- uses the sum
            float sacc = 0.f;
            if (rk_ok)
                for (int i = jl; i < A; i += 16) sacc += sb[rk * A + i];
            sacc = quad_sum(sacc);
            sacc += dpp_shift<0x124, 0xf>(sacc, 0.f);             // row_ror:4
            sacc += dpp_shift<0x128, 0xf>(sacc, 0.f);             // row_ror:8
            const float s = sacc;
            float grad = 0.f;
            if (MODE == MODE_PPO) {
                // v_exp_f32 on (s - lpo) * log2 e: |s - lpo| is O(0.1-1), so the product's rounding is far below the
                // instruction's own 1 ulp
                const float ratio = __builtin_amdgcn_exp2f((s - lpo) * PFPN_LOG2E);
                const float sur = ratio * nadv;
                const float cr = fmaxf(fminf(ratio, ppo.clip_hi), ppo.clip_lo);
                const float csur = cr * nadv;
                // grad of -mean(min(sur, csur)): tf.minimum routes to `sur` where sur <= csur,
                // clip_by_value passes where lo <= ratio <= hi.
                const bool pass = (sur <= csur) || (ratio >= ppo.clip_lo && ratio <= ppo.clip_hi);
                grad = pass ? (-ppo.inv_B * nadv) * ratio : 0.f;
                if (owner) {
                    pol_sum += fminf(sur, csur);
                    if (ppo.ratio_out) ppo.ratio_out[b] = ratio;
                }
            }
            if (owner) {
                __hip_atomic_store(&gslot[(k & 3) * R + rk],
                                   ((unsigned long long)__float_as_uint(grad) << 32) | (unsigned)(k + 1),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                logp_out[b] = s;
            }
            if (k < 4) PFPN_STAMP(5 + k * 6);
            if (++slot == P) { slot = 0; ++lap; }
        }
        if (MODE == MODE_PPO) {
            // loss sums (lanes 0, 16, 32, 48 hold them: rows in tile order)
            const int nq = ppo.want_db ? 3 : 2;
            float* slab = partial + (size_t)blockIdx.x * (nq * AN + 4);
            const float ps = wave_sum_dpp(pol_sum), vs = wave_sum_dpp(val_sum);      // lane 63 = total
            if (lane == 63) { slab[nq * AN + 0] = ps; slab[nq * AN + 1] = vs; }
        }
        return;
    }

    // =========================== storer waves ================================================
    if (NS && wave >= NC + TK_NL + NRW) {
        const int s_ = wave - (NC + TK_NL + NRW);
        constexpr int SU = 5;                              // float4 per lane and round: all reads of a round, then its stores
        int slot = 0, lap = 0;
        for (int k = 0; k + 1 < T; ++k) {                  // (the block's last tile is stored by the compute waves)
            const int row0 = (tile0 + k) * R, rows = min(R, B - row0);
            const int nv4 = rows * AN / 4;
            while (lds_counter(&bdone[slot]) < NC * (lap + 1)) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const f4v* src = reinterpret_cast<const f4v*>(tile + (size_t)slot * TS);
            f4v* dst = reinterpret_cast<f4v*>(dlogits + (size_t)row0 * AN);
            for (int c0 = s_; c0 * 64 < nv4; c0 += SU * NS) {
                // SU clamped, unconditional LDS reads in one round trip, then the stores.  The empty asm ties the five
                // values together: hipcc otherwise pairs each read with its store (one LDS round trip per KiB).
                const int i0 = c0 * 64 + lane, i1 = i0 + NS * 64, i2 = i1 + NS * 64, i3 = i2 + NS * 64, i4 = i3 + NS * 64;
                f4v v0 = src[min(i0, nv4 - 1)], v1 = src[min(i1, nv4 - 1)], v2 = src[min(i2, nv4 - 1)],
                    v3 = src[min(i3, nv4 - 1)], v4 = src[min(i4, nv4 - 1)];
                asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4));
                // non-temporal: the 20 MB of dlogits are not left dirty in L2 for the end-of-kernel write-back
                // (measured 13.8 -> 12.6 us per launch, and the step gains too although its next kernels read dlogits;
                // nt on the scattered slab stores cost +3 us, nt on the DMA loads changed nothing)
                if (i0 < nv4) __builtin_nontemporal_store(v0, &dst[i0]);
                if (i1 < nv4) __builtin_nontemporal_store(v1, &dst[i1]);
                if (i2 < nv4) __builtin_nontemporal_store(v2, &dst[i2]);
                if (i3 < nv4) __builtin_nontemporal_store(v3, &dst[i3]);
                if (i4 < nv4) __builtin_nontemporal_store(v4, &dst[i4]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of the plane have returned
            if (lane == 0) atomicAdd(&freed[slot], 1);
            if (k < 4) PFPN_STAMP(8 + k * 6);
            if (++slot == P) { slot = 0; ++lap; }
        }
        return;
    }

    // =========================== compute waves ===============================================
    // lane = 16 d + 4 r + g: dim 4*wave + d, row r of the tile, particles g, g+4, ...
    const int d = lane >> 4, r = (lane >> 2) & 3, g = lane & 3;
    const int a_raw = wave * 4 + d;
    const bool dim_ok = a_raw < A;
    const int a = dim_ok ? a_raw : A - 1;                  // clamped: addresses stay inside the plane
    const int sidx = r * A + a;                            // segment slot in the tile
    const int ndw = min(4, A - wave * 4);                  // my dims (per row)
    float* seg0 = tile + sidx * N + g;
    const bool last_ok = (g + TK_G * (NP - 1)) < N;       // does my NP-th particle exist?
    float b_loc[NP], b_isc[NP], b_ct[NP];
    float acc_loc[NP], acc_ls[NP], acc_db[NP];
    {
        // The tables of my wave's dims, transformed once per wave:
        //   scale = exp(logstd), cterm = 0.5log(2pi)+log(scale) as the reference graph computes them, pre-scaled
        //   to base 2:  z' = (x - loc) * isc2,  phi = exp2(-z'^2 - ct2),  isc2 = sqrt(0.5 log2 e)/scale
        //   +64: e*phi is formed as ONE exp2 of (softmax exponent + Gaussian exponent + 64), so products
        //   down to 2^-190 stay normal; the 2^-64 is taken back after the segment sum.
        const int e0 = wave * 4 * N, ne = ndw * N;
#pragma unroll
        for (int u = 0; u < BKU; ++u) {
            const int j = lane + 64 * u;
            if (j < ne) {
                const float sc = expf(bank_ls[u]);
                bank[e0 + j] = bank_lc[u];
                bank[AN + e0 + j] = (1.0f / sc) * PFPN_SQRT_HALF_LOG2E;
                bank[2 * AN + e0 + j] = 64.0f - (PFPN_HALF_LOG_2PI + logf(sc)) * PFPN_LOG2E;
            }
        }
        PFPN_STAMP(43);
        // same wave wrote what it now reads: a wave's LDS operations execute in issue order
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int n = min(g + TK_G * i, N - 1);
            b_loc[i] = bank[a * N + n];
            b_isc[i] = bank[AN + a * N + n];
            b_ct[i] = bank[2 * AN + a * N + n];
            acc_loc[i] = 0.f; acc_ls[i] = 0.f; acc_db[i] = 0.f;
        }
        // tile 0: my action values (and, backward mode, my rows' dlogp -- every wave writes the same four values) into
        // the planes the later tiles' DMAs fill; then my row pieces must have landed
        if (T > 0 && dim_ok && g == 0 && tile0 * R + r < B) xpl[sidx] = x0;
        if (MODE == MODE_BWD && T > 0 && d == 0 && g == 0 && tile0 * R + r < B) rpl[r] = g0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const float two_m64 = 5.421010862427522e-20f;      // 2^-64
    // The hardware issues a SIMD's oldest wave first, so the later a compute wave was launched the further it falls behind
    // -- and the block moves at the pace of its last wave (every row result needs all nine).  Inverted priorities
    // (waves 4-7: 1, wave 8 and up: 2) even the three generations out: 16.2 -> 15.9 us with the finalize launch, forward
    // 7.8 -> 7.4 us.
    if (wave >= 8) __builtin_amdgcn_s_setprio(2);
    else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    PFPN_STAMP(1);

    // tile k of this block lives in plane k % P; (slot, lap) = (k % P, k / P) are carried as counters
    auto wait_ready = [&](int slot, int lap) __attribute__((always_inline)) {
        const int target = TK_NL * (lap + 1);
        while (lds_counter(&ready[slot]) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    // forward of tile k: softmax numerators stay in the plane, per-segment log p is published to
    // segbuf[k mod 4] and the row arrival counters of that parity are bumped.
    auto forward = [&](int k, int slot, float& p_o, float& inv_o, v2f (&ef2_o)[NPK > 0 ? NPK : 1], float (&eft_o)[NP - 2 * NPK])
        __attribute__((always_inline)) {
        const int row0 = (tile0 + k) * R;
        const int rows = min(R, B - row0);
        const bool active = dim_ok && r < rows;
        float* seg = seg0 + slot * TS;
        // four log-p buffers: a wave that has run ahead writes tile k+2's (and, one iteration later, k+3's)
        // values while the row wave may still be summing tile k's; it cannot reach k+4 before the row wave
        // has finished k+2
        float* sb = segbuf + (k & 3) * nseg;
        float inv_sum = 0.f, p = 0.f;
        if (active) {
            const float xval = xpl[slot * XS + sidx];
            // particles in pairs (2k, 2k+1): v_pk_mul/add/fma_f32 give two results per issue slot; the last one or
            // two particles -- the only ones that may not exist -- stay scalar.  The maximum is taken on l*log2(e).
            v2f lp2[NPK > 0 ? NPK : 1];
            float lpt[NP - 2 * NPK];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < NPK; ++k) {
                lp2[k] = (v2f){seg[TK_G * 2 * k], seg[TK_G * (2 * k + 1)]} * PFPN_LOG2E;
                mx = fmaxf(fmaxf(lp2[k].x, lp2[k].y), mx);
            }
#pragma unroll
            for (int i = 2 * NPK; i < NP; ++i) {
                lpt[i - 2 * NPK] = 0.f;
                if (i < NP - 1 || last_ok) {
                    lpt[i - 2 * NPK] = seg[TK_G * i] * PFPN_LOG2E;
                    mx = fmaxf(mx, lpt[i - 2 * NPK]);
                }
            }
            mx = quad_max(mx);
            v2f s2 = {0.f, 0.f}, n2 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NPK; ++k) {
                // raw v_exp_f32 twice per particle: softmax numerator, and numerator * Normal.prob * 2^64 in one go
                const v2f le = lp2[k] - mx;
                const v2f z = (xval - (v2f){b_loc[2 * k], b_loc[2 * k + 1]}) * (v2f){b_isc[2 * k], b_isc[2 * k + 1]};
                const v2f arg = le + __builtin_elementwise_fma(-z, z, (v2f){b_ct[2 * k], b_ct[2 * k + 1]});
                const v2f en = {__builtin_amdgcn_exp2f(le.x), __builtin_amdgcn_exp2f(le.y)};
                if (MODE != MODE_FWD) { seg[TK_G * 2 * k] = en.x; seg[TK_G * (2 * k + 1)] = en.y; }
                ef2_o[k] = (v2f){__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                if (k == 0) { s2 = en; n2 = ef2_o[k]; }     // start from the first term, not from 0 + it
                else { s2 += en; n2 += ef2_o[k]; }
            }
            float sum = s2.x + s2.y, num = n2.x + n2.y;
#pragma unroll
            for (int i = 2 * NPK; i < NP; ++i) {
                if (i < NP - 1 || last_ok) {
                    const float le = lpt[i - 2 * NPK] - mx;
                    const float zq = (xval - b_loc[i]) * b_isc[i];
                    const float en = __builtin_amdgcn_exp2f(le);
                    if (MODE != MODE_FWD) seg[TK_G * i] = en;
                    sum += en;
                    eft_o[i - 2 * NPK] = __builtin_amdgcn_exp2f(le + fmaf(-zq, zq, b_ct[i]));
                    num += eft_o[i - 2 * NPK];
                }
            }
            sum = quad_sum(sum);
            num = quad_sum(num);
            inv_sum = __builtin_amdgcn_rcpf(sum);
            p = (num * two_m64) * inv_sum;
            if (p < 1.401298464324817e-45f) p = 0.f;   // below the smallest fp32 denormal: the reference's p is 0
            if (g == 0) {
                float lp = __builtin_amdgcn_logf(p) * 0.6931471805599453f;      // v_log_f32 is log2
                if (tanh_mode) lp -= tanh_log_det(xval);
                if (pmix_out) pmix_out[(size_t)row0 * A + sidx] = p;
                sb[sidx] = lp;
            }
        }
        p_o = p; inv_o = inv_sum;
        if (MODE != MODE_BWD) {
            // publish: my dims of every live row have arrived.  The wait orders the ds_write of log p before the
            // ds_add that announces it (the order in which the LDS unit retires a wave's plain writes and its atomics
            // is not something to rely on).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < rows) atomicAdd(&cnt[(k & 3) * R + lane], ndw);
        }
        if (MODE == MODE_FWD) {
            // forward mode never stores the plane: it is free as soon as my reads of it are done
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(&freed[slot], 1);
        }
    };

    // e_n * phi_n * 2^64 of my particles (the forward's numerator terms) stay in registers until the tile's backward:
    // 9 VGPRs per tile in flight instead of a v_exp_f32 per particle to re-form them (the compute waves are VALU-bound,
    // and 28 of their ~190 VALU instructions per tile were quarter-rate transcendentals)
    v2f ef2[NPK > 0 ? NPK : 1], ef2n[NPK > 0 ? NPK : 1];
    float eft[NP - 2 * NPK], eftn[NP - 2 * NPK];
#pragma unroll
    for (int i = 0; i < (NPK > 0 ? NPK : 1); ++i) { ef2[i] = (v2f){0.f, 0.f}; ef2n[i] = (v2f){0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < NP - 2 * NPK; ++i) { eft[i] = 0.f; eftn[i] = 0.f; }
    float p = 0.f, inv_sum = 0.f, p1 = 0.f, inv1 = 0.f;
    if (T > 0) forward(0, 0, p, inv_sum, ef2, eft);        // tile 0: fetched by this wave itself
    PFPN_STAMP(2);

    int slot = 0, slot_n = (P > 1) ? 1 : 0, lap_n = (P > 1) ? 0 : 1;       // plane of tile k | plane and lap of tile k+1
    for (int k = 0; k < T; ++k) {
        const int row0 = (tile0 + k) * R;
        const int rows = min(R, B - row0);
        float* seg = seg0 + slot * TS;
        const bool active = dim_ok && r < rows;
        if (k < 4) PFPN_STAMP(3 + k * 6);
        {   // ... and the three generations trade priorities tile by tile, so that none stays the block's laggard:
            // 15.9 -> 15.3-15.4 us with finalize, backward 14.8 -> 14.5 (A/B on one box, tools/mix_var.sh)
            const int pr = ((wave >> 2) + 2 * k) % 3;
            if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
        }

        // ---- forward of the next tile --------------------------------------------------------
        if (k + 1 < T) { wait_ready(slot_n, lap_n); forward(k + 1, slot_n, p1, inv1, ef2n, eftn); }
        if (k < 4) PFPN_STAMP(4 + k * 6);

        // ---- row stage of this tile: done by the row wave; wait for it and pick up my row's gradient ---------
        float my_g = 0.f;                               // d loss / d logp of my row
        if (MODE == MODE_BWD) {
            if (active) my_g = rpl[slot * 64 + r];
        } else {
            // (forward mode waits too: it keeps the compute waves within two tiles of the row wave, which is what
            // the four segment-log-p buffers allow)
            const unsigned long long* gs = &gslot[(k & 3) * R + (r < rows ? r : 0)];
            unsigned long long gv;
            do {
                gv = __hip_atomic_load(gs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } while ((unsigned)gv != (unsigned)(k + 1));
            asm volatile("" ::: "memory");
            if (MODE == MODE_PPO && active) my_g = __uint_as_float((unsigned)(gv >> 32));
        }
        if (k < 4) PFPN_STAMP(6 + k * 6);

        if (MODE != MODE_FWD) {
            // ---- backward on my quarter (dlogits overwrite the LDS slot) -----------------------
            if (active) {
                const float xv = xpl[slot * XS + sidx];
                const float q_ = guard_finite(my_g * __builtin_amdgcn_rcpf(p));
                const float qs = q_ * inv_sum;
                const float qp = qs * p;
                const float qs64 = qs * two_m64;
                float dxa = 0.f;
                v2f dx2 = {0.f, 0.f};
                // pairs of particles, as in the forward
#pragma unroll
                for (int k = 0; k < NPK; ++k) {
                    const v2f en = {seg[TK_G * 2 * k], seg[TK_G * (2 * k + 1)]};
                    const v2f isc = {b_isc[2 * k], b_isc[2 * k + 1]};
                    const v2f z = (xv - (v2f){b_loc[2 * k], b_loc[2 * k + 1]}) * isc;            // z / c
                    const v2f tt = ef2[k] * qs64;                                                  // q * w_n * phi_n
                    const v2f dlg = __builtin_elementwise_fma(-en, (v2f){qp, qp}, tt);            // q * w_n * (phi_n - p)
                    seg[TK_G * 2 * k] = dlg.x; seg[TK_G * (2 * k + 1)] = dlg.y;
                    if (MODE == MODE_PPO) { acc_db[2 * k] += dlg.x; acc_db[2 * k + 1] += dlg.y; }   // fc_policy bias gradient
                    const v2f al = __builtin_elementwise_fma(tt, z, (v2f){acc_loc[2 * k], acc_loc[2 * k + 1]});     // scaled by isc*c in the tail
                    acc_loc[2 * k] = al.x; acc_loc[2 * k + 1] = al.y;
                    const v2f as = __builtin_elementwise_fma(
                        tt, __builtin_elementwise_fma(z * z, (v2f){PFPN_2LN2, PFPN_2LN2}, (v2f){-1.f, -1.f}),
                        (v2f){acc_ls[2 * k], acc_ls[2 * k + 1]});                                 // tt*(z^2 - 1)
                    acc_ls[2 * k] = as.x; acc_ls[2 * k + 1] = as.y;
                    if (dx_out) dx2 = __builtin_elementwise_fma(tt * z, isc * PFPN_2LN2, dx2);     // isc*c = isc2*2ln2
                }
#pragma unroll
                for (int i = 2 * NPK; i < NP; ++i) {
                    if (i < NP - 1 || last_ok) {
                        const float en = seg[TK_G * i];
                        const float zq = (xv - b_loc[i]) * b_isc[i];
                        const float tt = qs64 * eft[i - 2 * NPK];
                        const float dlg = fmaf(-en, qp, tt);
                        seg[TK_G * i] = dlg;
                        if (MODE == MODE_PPO) acc_db[i] += dlg;
                        acc_loc[i] = fmaf(tt, zq, acc_loc[i]);
                        acc_ls[i] = fmaf(tt, fmaf(zq * zq, PFPN_2LN2, -1.f), acc_ls[i]);
                        if (dx_out) dxa = fmaf(tt * zq, b_isc[i] * PFPN_2LN2, dxa);
                    }
                }
                dxa += dx2.x + dx2.y;
                if (dx_out) {
                    dxa = quad_sum(dxa);
                    if (g == 0) {
                        float dd = -dxa;
                        if (tanh_mode) dd += my_g * 2.f * tanhf(xv);
                        dx_out[(size_t)row0 * A + sidx] = dd;
                    }
                }
            }
            if (k < 4) PFPN_STAMP(7 + k * 6);
            // ---- the plane now holds my part of dlogits: hand it to the storer waves -- except for the block's LAST
            //      tile, whose pieces every compute wave stores itself the moment its own backward is done: the storers
            //      would start on it only when the slowest wave has finished, with the whole block waiting (~2,000 cycles
            //      of the kernel's tail).  Four row pieces of ndw N floats, one float4 per lane, non-temporal.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my ds_writes are done (in-order LDS)
            if (k + 1 < T) {
                if (lane == 0) atomicAdd(&bdone[slot], 1);
            } else {
                const int nf4 = (ndw * N) >> 2;
                const f4v* src = reinterpret_cast<const f4v*>(tile + (size_t)slot * TS + wave * 4 * N) + lane;
                f4v* dst = reinterpret_cast<f4v*>(dlogits + (size_t)row0 * AN + wave * 4 * N) + lane;
                if (lane < nf4) {
                    f4v v[R];
#pragma unroll
                    for (int rr = 0; rr < R; ++rr) v[rr] = src[(size_t)min(rr, rows - 1) * (AN >> 2)];
                    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
                    for (int rr = 0; rr < R; ++rr)
                        if (rr < rows) __builtin_nontemporal_store(v[rr], &dst[(size_t)rr * (AN >> 2)]);
                }
            }
        }
        p = p1; inv_sum = inv1;
#pragma unroll
        for (int i = 0; i < (NPK > 0 ? NPK : 1); ++i) ef2[i] = ef2n[i];
#pragma unroll
        for (int i = 0; i < NP - 2 * NPK; ++i) eft[i] = eftn[i];
        slot = slot_n;
        if (++slot_n == P) { slot_n = 0; ++lap_n; }
    }

    if (MODE == MODE_FWD) return;
    PFPN_STAMP(40);
    // ---- block partials: dloc / dlogstd (/ dlogits column sums).  The four rows of a tile that share my (a, n) set
    //      are the lanes 4, 8, 12 further on in my 16-lane DPP row: two row_ror adds leave a total in each of the four
    //      (with its own, fixed association order).  The row-q lane stores quantity q (0 dloc, 1 dlogstd, 2 column
    //      sums): one store instruction per particle instead of three.  All lanes run the DPP steps (idle lanes hold
    //      zeros).  Slab layout [nq*AN + 4] per block.
    const int nq = (MODE == MODE_PPO && ppo.want_db) ? 3 : 2;
    float* slab = partial + (size_t)blockIdx.x * (nq * AN + 4);
    const bool writer = dim_ok && r < nq;
    float* my_slab = slab + r * AN + a * N + g;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float v0 = acc_loc[i] * (b_isc[i] * PFPN_2LN2);     // dloc *= c/scale
        float v1 = acc_ls[i];
        float v2 = acc_db[i];
        v0 += dpp_shift<0x124, 0xf>(v0, 0.f); v0 += dpp_shift<0x128, 0xf>(v0, 0.f);
        v1 += dpp_shift<0x124, 0xf>(v1, 0.f); v1 += dpp_shift<0x128, 0xf>(v1, 0.f);
        if (MODE == MODE_PPO) { v2 += dpp_shift<0x124, 0xf>(v2, 0.f); v2 += dpp_shift<0x128, 0xf>(v2, 0.f); }
        if (writer && (i < NP - 1 || last_ok)) my_slab[TK_G * i] = r == 0 ? v0 : r == 1 ? v1 : v2;
    }
    if (!NRW && wave == 0 && lane == 0) { slab[nq * AN + 0] = 0.f; slab[nq * AN + 1] = 0.f; }
    PFPN_STAMP(41);
}

// ---------------------------------------------------------------------------------------
// wave path (any N): wave (a, chunk c) walks rows b = c, c+C, ...
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_mix_wave(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,
    int B, int A, int N, int C, int tanh_mode,
    float* __restrict__ lpseg_out,           // [B,A] per-segment log p (FWD)
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // BWD
    float* __restrict__ dx_out,              // BWD optional
    float* __restrict__ partial)             // [C][2*A*N]  (BWD)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= A * C) return;
    const int a = wave % A;
    const int c = wave / A;
    const int AN = A * N;
    const float* locr = loc + a * N;
    const float* lsr = logstd + a * N;
    // dloc/dlogstd accumulators for particles lane, lane+64, ...  (written per 64-chunk below)
    for (int n0 = 0; n0 < N; n0 += 64) {
        if (MODE == MODE_BWD && n0 + lane < N) {
            partial[(size_t)c * 2 * AN + a * N + n0 + lane] = 0.f;
            partial[(size_t)c * 2 * AN + AN + a * N + n0 + lane] = 0.f;
        }
    }
    for (int b = c; b < B; b += C) {
        const float* row = logits + (size_t)b * AN + a * N;
        const float xv = x[(size_t)b * A + a];
        float mx = -INFINITY;
        for (int n = lane; n < N; n += 64) mx = fmaxf(mx, row[n]);
        mx = wave_max(mx);
        float sum = 0.f, num = 0.f;
        for (int n = lane; n < N; n += 64) {
            const float sc = expf(lsr[n]);
            const float en = expf(row[n] - mx);
            const float z = (xv - locr[n]) * (1.0f / sc);
            const float ph = expf(fmaf(-0.5f * z, z, -(PFPN_HALF_LOG_2PI + logf(sc))));
            sum += en;
            num = fmaf(en, ph, num);
        }
        sum = wave_sum(sum);
        num = wave_sum(num);
        const float inv_sum = 1.0f / sum;
        const float p = num * inv_sum;
        if (MODE == MODE_FWD) {
            if (lane == 0) {
                float lp = logf(p);
                if (tanh_mode) lp -= tanh_log_det(xv);
                lpseg_out[(size_t)b * A + a] = lp;
                if (pmix_out) pmix_out[(size_t)b * A + a] = p;
            }
        } else {
            const float g = dlogp_in[b];
            const float q = guard_finite(g / p);
            const float qs = q * inv_sum;
            float dxa = 0.f;
            float* drow = dlogits + (size_t)b * AN + a * N;
            for (int n = lane; n < N; n += 64) {
                const float sc = expf(lsr[n]);
                const float isc = 1.0f / sc;
                const float en = expf(row[n] - mx);
                const float z = (xv - locr[n]) * isc;
                const float ph = expf(fmaf(-0.5f * z, z, -(PFPN_HALF_LOG_2PI + logf(sc))));
                const float tt = qs * en * ph;
                drow[n] = tt - (qs * en) * p;
                const float zi = z * isc;
                partial[(size_t)c * 2 * AN + a * N + n] += tt * zi;
                partial[(size_t)c * 2 * AN + AN + a * N + n] += tt * fmaf(z, z, -1.f);
                dxa = fmaf(tt, zi, dxa);
            }
            if (dx_out) {
                dxa = wave_sum(dxa);
                if (lane == 0) {
                    float d = -dxa;
                    if (tanh_mode) d += g * 2.f * tanhf(xv);
                    dx_out[(size_t)b * A + a] = d;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// wide path (N % 4 == 0, N >= 64; the 128 x 256 stress shape): one wavefront per segment, a lane
// owns M float4s (4 consecutive particles each): a segment is fetched as whole 1-KiB coalesced
// wave-loads straight into registers (no LDS).  Wave (a, c) keeps its action dim, so the particle
// bank and the dloc/dlogstd accumulators of its 4*M particles stay in registers while it walks rows
// b = c, c+C, ...; U rows are in flight per wave (U float4 loads per lane outstanding) to cover the
// HBM latency.  Reductions over a segment are wave shuffles.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// The wide kernel's read-once logits stream (1 GiB at 8192 x 128 x 256) takes the non-temporal cache policy in the
// FORWARD kernel: measured 207 -> 195 us there (0.65 -> 0.69 of 8 TB/s, two boxes, A/B in one session).  The backward
// kernel also streams 1 GiB of stores; STREAM = nt loads + nt stores there.  With the launch sized for ~12 waves per CU
// it changes nothing (428-437 us either way), but with nt stores the best launch moves to 8 waves per CU = exactly two
// blocks per CU (C = 16 at A = 128): 385-404 us against 430-437 for the default policy at its own best C, alternating
// builds over four rounds on one box (tools/wide_sweep.sh, profiles/r03_wide_stream.txt).  C = 18 (2.25 blocks per CU,
// an uneven last round) measures 460-470 us: the chunk count is kept to a whole number of blocks per CU.  STREAM is
// only taken when the dlogits tensor is larger than the 256 MiB infinity cache: below that the dW GEMM that follows
// re-reads dlogits from cache, which a streaming store would give up.
template <bool NT>
__device__ __forceinline__ float4 ld4_stream(const float* p) {
    if (NT) {
#if defined(PFPN_WIDE_LD_POLICY) && PFPN_WIDE_LD_POLICY == 1          // A/B build knobs (tools/ab_build.sh)
        return ld4(p);
#elif defined(PFPN_WIDE_LD_POLICY) && PFPN_WIDE_LD_POLICY == 2
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));      // (placeholder: same as default)
        return make_float4(v.x, v.y, v.z, v.w);
#else
        const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
#endif
    }
    return ld4(p);
}
// Streaming stores of the backward's dlogits (round 4): WRITE-THROUGH (`sc1`: the line leaves the XCD's L2 for memory and is
// dropped) instead of non-temporal -- 384 us against 394 us for the 2 GiB pass on one box, a tie (389 / 391) on another, in
// alternating runs (profiles/r04_wide_handoff_ab.txt; `sc0 sc1` the same, `sc0 sc1 nt` 388).  Issued as a raw BUFFER store
// whose resource descriptor is the (wave-uniform) row segment: the builtin takes the cache policy as an argument (16 = sc1 on
// gfx940+) and is visible to hipcc's waitcnt pass (an asm store is not; a volatile store, `sc0 sc1`, makes hipcc drain
// vmcnt(0) in front of every store).
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef float f3v __attribute__((ext_vector_type(3)));
template <bool NT>
__device__ __forceinline__ void st4_row(float* row, int seg_bytes, int q, float a, float b, float c, float d) {
    if (NT) {
        const f4v v = {a, b, c, d};
#if defined(PFPN_WIDE_ST_POLICY) && PFPN_WIDE_ST_POLICY == 0           // A/B build knobs (tools/ab_build.sh)
        __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(row + 4 * q));
#elif defined(PFPN_WIDE_ST_POLICY) && PFPN_WIDE_ST_POLICY == 1
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(row + 4 * q), "v"(v) : "memory");
#else
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row, 0, seg_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rsrc, 16 * q, 0, 16);
#endif
        return;
    }
    *reinterpret_cast<float4*>(row + 4 * q) = make_float4(a, b, c, d);
}

template <bool V> struct BoolTag { static constexpr bool value = V; };

// ---- in-launch hand-off of the wide kernels' partial results (round 4: no separate row-sum / finalize launch) ----------
// Producers publish their partials with WRITE-THROUGH stores (`sc1`: the bytes leave the XCD's L2 for memory), drain them
// (`s_waitcnt vmcnt(0)`), then take a ticket from an agent-scope counter; whoever draws the LAST ticket of its group reads
// every partial with `sc1` loads (served past its CU's L1 and its XCD's L2: another XCD's L2 is not coherent with it) and
// adds them in a FIXED order -- bitwise the result the separate launch gave.  No fence: an agent-scope release would write
// back this XCD's whole L2, which holds the kernel's 1 GiB dlogits stream (MI355X_MICROARCH.md, inter-workgroup
// visibility: `sc1` payload -> vmcnt(0) -> atomic ticket -> `sc1` loads by the wave whose add returned last).
// The counters live in the CALLER'S workspace (its last WIDE_TICK_WORDS words, see wide_tickets()): two launches in flight on two
// streams with two workspaces share nothing.  The entry point zeroes the counters a launch uses stream-ordered in front of it
// (hipMemsetAsync: a memset node under capture), so a fresh (uninitialised) or poisoned workspace is repaired, not trusted.
constexpr int WIDE_TICK_FWD = 0;                     // [64]   one per row chunk c (C <= 64)
constexpr int WIDE_TICK_PPO = 64;                    // [1]    the chunks whose PPO head is done (forward + PPO head in one launch)
constexpr int WIDE_TICK_BWD = 128;                   // [4096] one per action dim
constexpr int WIDE_FOLD_MAX_A = 4096;
constexpr int WIDE_FOLD_MAX_LDS = 32 * 1024;
constexpr size_t WIDE_TICK_WORDS = WIDE_TICK_BWD + WIDE_FOLD_MAX_A;
// pfpn_ppo_wide_fused's exchange area, in front of the tickets: [C <= 32][4 slots][nb <= 32][8] tagged partials, [C][nb] XCC ids,
// [C][2] loss sums
constexpr size_t WIDE_F2L_PART_WORDS = (size_t)32 * 4 * 32 * 8, WIDE_F2L_XCC_WORDS = 32 * 32;
constexpr size_t WIDE_F2L_SYNC_WORDS = WIDE_F2L_PART_WORDS + WIDE_F2L_XCC_WORDS + 64;

__device__ __forceinline__ void st4_sc1(float* p, float a, float b, float c, float d) {
    const f4v v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// eight `sc1` loads and their wait as ONE asm statement: the compiler does not see these loads complete, so nothing of its
// own (a copy at a control-flow join, a spill) may sit between a load and the wait -- the first version, one asm per load
// with a conditional around it, summed stale registers.  Loads past `n` re-read entry n - 1 (always a valid address) and
// are ignored by the caller.  Early-clobber outputs: a result must not land in a register a later load's address lives in.
__device__ __forceinline__ void ld4x8_sc1(const float* p, size_t stride, int n, f4v (&v)[8]) {
    const float* q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = p + (size_t)min(i, n - 1) * stride;
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %9, off sc1\n\t"
        "global_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\t"
        "global_load_dwordx4 %4, %12, off sc1\n\t"
        "global_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\t"
        "global_load_dwordx4 %7, %15, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
        : "memory");
}

// The clipped-PPO head (k_ppo_loss's arithmetic, reference ppo.py:31-54) for the rows a chunk's last block has just finished:
// dlogp / dvalue per row, the two loss sums per chunk; the chunk that finishes last adds the C pairs in chunk order.
struct WidePpo {
    const float* logp_old; const float* adv; const float* value; const float* value_old; const float* stats;
    float clip_lo, clip_hi, value_coef, inv_B;
    float* dlogp; float* dvalue; float* loss3;
    float* sums;                                      // [C][2] in the workspace
};

// FOLD (forward): the row sum over the A segments happens inside the launch -- a wave keeps its rows' log p in LDS, the
// block (4 consecutive action dims of ONE row chunk: A % 4 == 0) adds its four waves' values per row and publishes ONE
// partial per row ([C][A/4][R4] in `lpseg_out`, whole float4s), the last block of a chunk adds the A/4 partials in block
// order into fin0 = logp[B].  The loop then holds no global store at all.  Backward: `fin0` / `fin1` != nullptr = the last
// of the C waves of an action dim adds the C slabs (k_finalize's order) into dloc / dlogstd.
template <int M, int MODE, bool TANH, bool FULLN, bool STREAM = false, bool FOLD = false>
__global__ __launch_bounds__(256) void k_mix_wide(
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    const float* __restrict__ dlogp_in,
    int B, int A, int N, int C,
    float* __restrict__ lpseg_out,           // [B,A] per-segment log p (FWD); FOLD: the per-block partials
    float* __restrict__ pmix_out,            // [B,A] optional
    float* __restrict__ dlogits,             // BWD
    float* __restrict__ dx_out,              // BWD optional
    float* __restrict__ partial,             // [C][2*A*N]  (BWD)
    float* __restrict__ fin0 = nullptr,      // FWD FOLD: logp [B];  BWD: dloc [A,N] (null: slabs left for the caller)
    float* __restrict__ fin1 = nullptr,      // BWD: dlogstd [A,N]
    WidePpo ppo = WidePpo{},                 // FWD FOLD: logp_old != nullptr -> the PPO head runs in the hand-off tail
    unsigned* __restrict__ tick = nullptr)   // the hand-off tickets (caller's workspace, zeroed in front of the launch)
{
    constexpr int U = 4;                                    // rows in flight per wave (reduced together)
    constexpr bool NT_LOADS = STREAM;                       // (forward: STREAM = the logits are read once and not again soon)
    extern __shared__ __align__(16) float s_lp[];           // FOLD: [4 waves][R4] log p of this block's rows
    const int lane = threadIdx.x & 63;
    // wave-uniform indices through readfirstlane: row/segment address arithmetic goes to the scalar unit
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= A * C) return;                              // (never taken with FOLD: the grid is exactly A * C / 4 blocks)
    const int R4 = ((B + C - 1) / C + 3) & ~3;              // rows per chunk, padded to whole float4s
    const int a = wave % A;                                 // neighbouring waves -> neighbouring dims
    const int c = wave / A;
    const int AN = A * N;
    const int nq = N >> 2;                                  // float4s per segment
    // bank in registers: loc, isc2 = sqrt(.5 log2 e)/scale, ct64 = 64 - (0.5log2pi + log scale)*log2 e.
    // e*phi = exp2(l*log2e - max*log2e - z'^2 + ct64) * 2^-64 in ONE v_exp_f32: the +64 keeps every
    // product that matters far above the flush-to-zero threshold of the raw instruction.
    bool ok[M];
    v2f b_loc2[M][2], b_isc2[M][2], b_ct2[M][2];            // float2 halves (xy | zw) of each lane's float4
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int qi = lane + 64 * j;
        ok[j] = FULLN || qi < nq;                            // FULLN: N == 256*M, every lane owns particles
        const int o = a * N + 4 * (ok[j] ? qi : 0);
        const float4 lo = ld4(loc + o), ls = ld4(logstd + o);
        const float lo_[4] = {lo.x, lo.y, lo.z, lo.w}, ls_[4] = {ls.x, ls.y, ls.z, ls.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = expf(ls_[k]);
            b_loc2[j][k >> 1][k & 1] = lo_[k];
            b_isc2[j][k >> 1][k & 1] = (1.0f / sc) * PFPN_SQRT_HALF_LOG2E;
            b_ct2[j][k >> 1][k & 1] = 64.f - (PFPN_HALF_LOG_2PI + logf(sc)) * PFPN_LOG2E;
        }
    }
    // BWD + FOLD = the backward that ALSO leaves the column sums of dlogits (the head's bias gradient: a third plane of the
    // slabs, finished into `lpseg_out`, which the backward does not otherwise use) -- one register set more, and the
    // 134-MB pass that re-read dlogits for it is gone (config 5's 1024-row shard: 21 us)
    constexpr bool DBIAS = MODE == MODE_BWD && FOLD;
    constexpr int PL = DBIAS ? 3 : 2;                       // planes per slab
    v2f acc_loc2[M][2], acc_ls2[M][2], acc_b2[M][2];
#pragma unroll
    for (int j = 0; j < M; ++j)
        for (int h = 0; h < 2; ++h) {
            acc_loc2[j][h] = (v2f){0.f, 0.f}; acc_ls2[j][h] = (v2f){0.f, 0.f}; acc_b2[j][h] = (v2f){0.f, 0.f};
        }
    const float two_m64 = 5.421010862427522e-20f;           // 2^-64

    // One group of U rows.  FULL = every row of the group exists: the body is then BRANCH-FREE straight-line
    // code (stores unconditional), so hipcc's waitcnt pass can count the stores younger than the prefetch
    // loads and emit s_waitcnt vmcnt(k) instead of vmcnt(0) -- otherwise every iteration would also wait for
    // its own stores to be acknowledged (~1 us exposed per iteration; CDNA4 vmcnt counts stores).
    // The per-particle arithmetic runs on float2 halves of each float4 (v_pk_mul/add/fma_f32: two fp32 results per
    // issue slot) -- at 128 x 256 this kernel is VALU-issue bound, not HBM bound: 17 VALU slots per particle
    // before, ~10 now, 8 of them the two quarter-rate v_exp_f32.  The row maximum is taken on l*log2(e) (a product:
    // no canonicalisation instructions in front of v_max3_f32, unlike on raw loads).
    auto group = [&](int b0, int gi, const float4 (&l)[U][M], const float (&xv)[U], const float (&gv)[U], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        float mx[U];
        v2f lp[U][M][2];                                      // l * log2(e)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mx[u] = -INFINITY;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                lp[u][j][0] = (v2f){l[u][j].x, l[u][j].y} * PFPN_LOG2E;
                lp[u][j][1] = (v2f){l[u][j].z, l[u][j].w} * PFPN_LOG2E;
                if (ok[j]) {                                  // two v_max3_f32 per float4
                    mx[u] = fmaxf(fmaxf(lp[u][j][0].x, lp[u][j][0].y), mx[u]);
                    mx[u] = fmaxf(fmaxf(lp[u][j][1].x, lp[u][j][1].y), mx[u]);
                }
            }
        }
        wave_reduce4<true>(mx[0], mx[1], mx[2], mx[3], lane);
        v2f e[U][M][2], ef[U][M][2], zq[U][M][2];
        float sum[U], num[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2f s2 = {0.f, 0.f}, n2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < M; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    e[u][j][h] = ef[u][j][h] = zq[u][j][h] = (v2f){0.f, 0.f};
                    if (ok[j]) {
                        // (the maximum on the raw logits + l c - max c as one packed fma -- 8 VALU instructions fewer per
                        // group -- measured no faster: 176 vs 177 us, round 4)
                        const v2f le = lp[u][j][h] - mx[u];
                        const v2f z = (xv[u] - b_loc2[j][h]) * b_isc2[j][h];
                        const v2f arg = le + __builtin_elementwise_fma(-z, z, b_ct2[j][h]);
                        zq[u][j][h] = z;
                        e[u][j][h] = (v2f){__builtin_amdgcn_exp2f(le.x), __builtin_amdgcn_exp2f(le.y)};
                        ef[u][j][h] = (v2f){__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                        if (j == 0 && h == 0) {              // start from the first term, not from 0 + it (a wasted v_pk_add)
                            s2 = e[u][j][h];
                            n2 = ef[u][j][h];
                        } else {
                            s2 += e[u][j][h];
                            n2 += ef[u][j][h];
                        }
                    }
                }
            }
            sum[u] = s2.x + s2.y;
            num[u] = n2.x + n2.y;
        }
        if (MODE == MODE_FWD) {
            // forward: nothing after the reductions needs a row's totals in every lane, so they stay distributed (lane l
            // holds row ur = wave_reduce4_row(l)) and the per-row tail -- v_rcp, v_log (quarter rate), the denormal
            // guard, the store -- runs ONCE across the four rows instead of four times on wave-uniform values
            const float ks = wave_reduce4_dist<false>(sum[0], sum[1], sum[2], sum[3], lane);
            const float kn = wave_reduce4_dist<false>(num[0], num[1], num[2], num[3], lane);
            const int ur = wave_reduce4_row(lane);
            const int b = b0 + ur * C;
            float p = (kn * two_m64) * __builtin_amdgcn_rcpf(ks);
            if (p < 1.401298464324817e-45f) p = 0.f;         // below the smallest fp32 denormal: the reference's p is 0
            float lpr = __builtin_amdgcn_logf(p) * 0.6931471805599453f;     // v_log_f32 is log2
            if (TANH) {
                const float xr = ur == 0 ? xv[0] : ur == 1 ? xv[1] : ur == 2 ? xv[2] : xv[3];
                lpr -= tanh_log_det(xr);
            }
            if (FOLD) {
                // row c + (4 gi + ur) C of this chunk: kept in LDS until the block's hand-off (rows past B in the ragged
                // last group are clamped duplicates: finite, never read back)
                if ((lane & 15) == 0) s_lp[(threadIdx.x >> 6) * R4 + 4 * gi + ur] = lpr;
                if (pmix_out && (FULL || b < B)) pmix_out[(size_t)b * A + a] = p;
                return;
            }
            // 16 lanes per row store the same value to the same address: one instruction, four transactions, no branch
            if (FULL || b < B) {
                lpseg_out[(size_t)b * A + a] = lpr;
                if (pmix_out) pmix_out[(size_t)b * A + a] = p;
            }
            return;
        }
        wave_reduce4<false>(sum[0], sum[1], sum[2], sum[3], lane);
        wave_reduce4<false>(num[0], num[1], num[2], num[3], lane);
        float dxa[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + u * C;
            const bool live = FULL || b < B;
            const float inv_sum = __builtin_amdgcn_rcpf(sum[u]);    // wave-uniform: one v_rcp, not an IEEE divide
            float p = (num[u] * two_m64) * inv_sum;
            if (p < 1.401298464324817e-45f) p = 0.f;         // below the smallest fp32 denormal: the reference's p is 0
            dxa[u] = 0.f;
            if (MODE == MODE_FWD) {
                float lpr = __builtin_amdgcn_logf(p) * 0.6931471805599453f;     // v_log_f32 is log2
                if (TANH) lpr -= tanh_log_det(xv[u]);
                // every lane stores the same value to the same address (one transaction, no branch)
                if (live) {
                    lpseg_out[(size_t)b * A + a] = lpr;
                    if (pmix_out) pmix_out[(size_t)b * A + a] = p;
                }
            } else {
                const float q_ = live ? guard_finite(gv[u] * __builtin_amdgcn_rcpf(p)) : 0.f;
                const float qs = q_ * inv_sum;
                const float qp = qs * p;
                const float qs64 = qs * two_m64;
                float* drow = dlogits + (size_t)b * AN + (size_t)a * N;
                v2f dx2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    if (ok[j]) {
                        v2f d[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const v2f tt = ef[u][j][h] * qs64;
                            const v2f z = zq[u][j][h];
                            d[h] = __builtin_elementwise_fma(-e[u][j][h], (v2f){qp, qp}, tt);
                            if (DBIAS) acc_b2[j][h] += d[h];                 // (q = 0 on rows past B: nothing added)
                            acc_loc2[j][h] = __builtin_elementwise_fma(tt, z, acc_loc2[j][h]);
                            acc_ls2[j][h] = __builtin_elementwise_fma(
                                tt, __builtin_elementwise_fma(z * z, (v2f){PFPN_2LN2, PFPN_2LN2}, (v2f){-1.f, -1.f}),
                                acc_ls2[j][h]);
                            if (dx_out) dx2 = __builtin_elementwise_fma(tt * z, b_isc2[j][h] * PFPN_2LN2, dx2);
                        }
                        if (live) st4_row<STREAM>(drow, 4 * N, lane + 64 * j, d[0].x, d[0].y, d[1].x, d[1].y);
                    }
                }
                dxa[u] = dx2.x + dx2.y;
            }
        }
        if (MODE == MODE_BWD && dx_out) {
            wave_reduce4<false>(dxa[0], dxa[1], dxa[2], dxa[3], lane);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + u * C;
                if (FULL || b < B) {
                    float dd = -dxa[u];
                    if (TANH) dd += gv[u] * 2.f * tanhf(xv[u]);
                    dx_out[(size_t)b * A + a] = dd;
                }
            }
        }
    };

    // Register double buffering: the loads of the NEXT group are issued before the current group is
    // processed, so a wave's own compute overlaps its HBM latency instead of adding to it.
    // (Two named register sets taking turns -- the body written out twice, no v_mov per loaded register -- was measured:
    // forward unchanged, backward 388 -> 414 us: the exit test between the two bodies costs the counted vmcnt waits.)
    float4 ln[U][M];
    float xn[U], gn[U];
#define PFPN_WIDE_LOAD_INTO(L_, X_, G_, b0_)                                                       \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                \
        const int b_ = min((b0_) + u * C, B - 1);            /* clamped: loads unconditional */     \
        const float* row_ = logits + (size_t)b_ * AN + (size_t)a * N;                              \
        _Pragma("unroll") for (int j = 0; j < M; ++j) L_[u][j] = ld4_stream<NT_LOADS>(row_ + 4 * (ok[j] ? lane + 64 * j : 0)); \
        X_[u] = x[(size_t)b_ * A + a];                                                             \
        G_[u] = (MODE == MODE_BWD) ? dlogp_in[b_] : 0.f;                                           \
    }
#define PFPN_WIDE_LOAD(b0_) PFPN_WIDE_LOAD_INTO(ln, xn, gn, b0_)
    const int step = C * U;
    int b0 = c, gi = 0;
    PFPN_WIDE_LOAD(b0)
    auto full_group = [&]() {                                // one full group: branch-free body
        float4 l[U][M];
        float xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < M; ++j) l[u][j] = ln[u][j];
            xv[u] = xn[u];
            gv[u] = gn[u];
        }
        PFPN_WIDE_LOAD(b0 + step)                            // clamped rows: harmless beyond the end
        group(b0, gi, l, xv, gv, BoolTag<true>{});
    };
    // Backward: the FIRST group runs outside the loop and the loop sits INSIDE that branch.  hipcc's waitcnt pass merges the
    // states of a loop header's predecessors conservatively: coming from the prologue only the prefetch loads are outstanding,
    // coming round the loop the loads AND the younger stores are -- merged, the header waited vmcnt(3..0) for its rows, i.e.
    // every iteration also waited for the previous group's stores to be acknowledged.  This way both predecessors end in
    // [loads, stores] and the header waits vmcnt(6..5): the stores stay in flight across the iteration (CDNA4 counts stores
    // in vmcnt).  388 us against 395 us (alternating, three rounds) with write-through stores; with non-temporal stores the
    // counted form is SLOWER (417-427 us).  -DPFPN_WIDE_NO_PEEL: the plain loop (A/B).
#if !defined(PFPN_WIDE_NO_PEEL)
    if (MODE == MODE_BWD) {
        if (b0 + (U - 1) * C < B) {
            full_group();
            b0 += step; ++gi;
            while (b0 + (U - 1) * C < B) { full_group(); b0 += step; ++gi; }
        }
    } else
#endif
    for (; b0 + (U - 1) * C < B; b0 += step, ++gi) full_group();
    if (b0 < B) group(b0, gi, ln, xn, gn, BoolTag<false>{}); // ragged tail (1..U-1 live rows)
#undef PFPN_WIDE_LOAD
#undef PFPN_WIDE_LOAD_INTO
    if (MODE == MODE_FWD && FOLD) {
        // ---- hand-off of the row sums (see the helpers above) ----
        __shared__ unsigned s_ticket;
        const int nb = A >> 2;                               // blocks per row chunk
        const int jb = blockIdx.x % nb;                      // this block's place in its chunk (c = blockIdx.x / nb)
        const int Rc = (B - c + C - 1) / C;                  // rows of chunk c
        const int nq4 = (Rc + 3) >> 2;
        float* part = lpseg_out + ((size_t)c * nb + jb) * R4;
        __syncthreads();
        for (int t = threadIdx.x; t < nq4; t += blockDim.x) {
            const float4 w0 = *reinterpret_cast<const float4*>(s_lp + 4 * t), w1 = *reinterpret_cast<const float4*>(s_lp + R4 + 4 * t);
            const float4 w2 = *reinterpret_cast<const float4*>(s_lp + 2 * R4 + 4 * t), w3 = *reinterpret_cast<const float4*>(s_lp + 3 * R4 + 4 * t);
            st4_sc1(part + 4 * t, ((w0.x + w1.x) + w2.x) + w3.x, ((w0.y + w1.y) + w2.y) + w3.y,
                    ((w0.z + w1.z) + w2.z) + w3.z, ((w0.w + w1.w) + w2.w) + w3.w);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            s_ticket = __hip_atomic_fetch_add(&tick[WIDE_TICK_FWD + c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != (unsigned)(nb - 1)) return;
        // the chunk's last block: logp[row] = sum over the chunk's blocks, in block order
        const float* all = lpseg_out + (size_t)c * nb * R4;
        const bool head = ppo.logp_old != nullptr;           // block-uniform (kernel argument)
        float hm = 0.f, hr = 1.f, hlo = 0.f, hhi = 0.f, ps = 0.f, vs = 0.f;
        if (head) { hm = ppo.stats[0]; hr = ppo.stats[1]; hlo = ppo.stats[2]; hhi = ppo.stats[3]; }
        for (int t = threadIdx.x; t < nq4; t += blockDim.x) {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
            for (int j0 = 0; j0 < nb; j0 += 8) {
                f4v v[8];
                ld4x8_sc1(all + (size_t)j0 * R4 + 4 * t, (size_t)R4, min(8, nb - j0), v);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (j0 + i < nb) acc = (j0 + i == 0) ? v[i] : acc + v[i];
            }
            const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = c + (4 * t + i) * C;
                if (b >= B) continue;
                fin0[b] = r[i];
                if (head) {                                  // k_ppo_loss, term for term
                    const float av = ppo.adv[b];
                    const float nadv = clip_adv((av - hm) * hr, hlo, hhi);
                    const float ratio = expf(r[i] - ppo.logp_old[b]);
                    const float sur = ratio * nadv;
                    const float csur = fmaxf(fminf(ratio, ppo.clip_hi), ppo.clip_lo) * nadv;
                    ps += fminf(sur, csur);
                    const bool pass = (sur <= csur) || (ratio >= ppo.clip_lo && ratio <= ppo.clip_hi);
                    ppo.dlogp[b] = pass ? (-ppo.inv_B * nadv) * ratio : 0.f;
                    const float dvv = ppo.value[b] - (av + ppo.value_old[b]);
                    vs += dvv * dvv;
                    ppo.dvalue[b] = 2.f * dvv * ppo.value_coef * ppo.inv_B;
                }
            }
        }
        if (threadIdx.x == 0) __hip_atomic_store(&tick[WIDE_TICK_FWD + c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (head) {
            __shared__ float hred[4];
            ps = block_sum(ps, hred);
            vs = block_sum(vs, hred);
            if (threadIdx.x == 0) {
                __hip_atomic_store(ppo.sums + 2 * c, ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (sc1 stores)
                __hip_atomic_store(ppo.sums + 2 * c + 1, vs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (__hip_atomic_fetch_add(&tick[WIDE_TICK_PPO], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(C - 1)) {
                    float tp = 0.f, tv = 0.f;                // the last chunk: the C pairs in chunk order
                    for (int k = 0; k < C; ++k) {
                        tp += __hip_atomic_load(ppo.sums + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tv += __hip_atomic_load(ppo.sums + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const float pl = -tp * ppo.inv_B, vl = tv * ppo.inv_B * ppo.value_coef;
                    ppo.loss3[0] = pl + vl; ppo.loss3[1] = pl; ppo.loss3[2] = vl;
                    __hip_atomic_store(&tick[WIDE_TICK_PPO], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (MODE == MODE_BWD) {
        float* slab = partial + (size_t)c * PL * AN + (size_t)a * N;
        const bool fold = fin0 != nullptr;                   // wave-uniform (kernel argument)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (ok[j]) {
                const v2f d0 = acc_loc2[j][0] * (b_isc2[j][0] * PFPN_2LN2), d1 = acc_loc2[j][1] * (b_isc2[j][1] * PFPN_2LN2);
                if (fold) {
                    st4_sc1(slab + 4 * (lane + 64 * j), d0.x, d0.y, d1.x, d1.y);
                    st4_sc1(slab + AN + 4 * (lane + 64 * j), acc_ls2[j][0].x, acc_ls2[j][0].y, acc_ls2[j][1].x, acc_ls2[j][1].y);
                    if (DBIAS)
                        st4_sc1(slab + 2 * AN + 4 * (lane + 64 * j), acc_b2[j][0].x, acc_b2[j][0].y, acc_b2[j][1].x, acc_b2[j][1].y);
                } else {
                    *reinterpret_cast<float4*>(slab + 4 * (lane + 64 * j)) = make_float4(d0.x, d0.y, d1.x, d1.y);
                    *reinterpret_cast<float4*>(slab + AN + 4 * (lane + 64 * j)) =
                        make_float4(acc_ls2[j][0].x, acc_ls2[j][0].y, acc_ls2[j][1].x, acc_ls2[j][1].y);
                }
            }
        }
        if (!fold) return;
        // ---- hand-off: the last of this action dim's C waves adds the C slabs, in k_finalize's order (16 running sums over
        // k = r, r + 16, ..., then those in order) so that the result is bitwise the one of the separate launch ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(&tick[WIDE_TICK_BWD + a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __builtin_amdgcn_readfirstlane(ticket);
        if (ticket != (unsigned)(C - 1)) return;
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) {
            const float* src = partial + (size_t)pl * AN + (size_t)a * N;
            float* dst = (pl == 0 ? fin0 : pl == 1 ? fin1 : lpseg_out) + (size_t)a * N;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if (!ok[j]) continue;
                f4v run[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) run[r] = (f4v){0.f, 0.f, 0.f, 0.f};
                for (int k0 = 0; k0 < C; k0 += 16) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f4v v[8];
                        const int kk = k0 + 8 * h;
                        if (kk >= C) break;
                        ld4x8_sc1(src + (size_t)kk * PL * AN + 4 * (lane + 64 * j), (size_t)PL * AN, min(8, C - kk), v);
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            if (kk + i < C) run[8 * h + i] += v[i];
                    }
                }
                f4v sum = run[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) sum += run[r];
                *reinterpret_cast<float4*>(dst + 4 * (lane + 64 * j)) = make_float4(sum.x, sum.y, sum.z, sum.w);
            }
        }
        if (lane == 0) __hip_atomic_store(&tick[WIDE_TICK_BWD + a], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// =====================================================================================================================
// The clipped-PPO learner pass of the 256-particle shapes as ONE launch that reads the logits once and writes dlogits once
// (2L; round 5).  The composition [forward + PPO head] -> backward reads the logits twice (3L): 598-617 us for 8192 x 128 x 256;
// this kernel: 379-390 us = 0.70-0.71 of 8 TB/s on its own 2.15 GB (tools/f2l_probe.py, profiles/r05_wide_2l_probe.txt).
//
// The obstacle is the row sum: log p of a row = the sum over its A action dims, and the wide kernels give every action dim to
// its own wave (bank + gradient accumulators of ONE dim in registers while it walks the rows): A = 128 waves on 32 blocks.  Here
// the waves of a row chunk MEET once per group of four rows, through memory, without atomics and without fences:
//   * a block's four waves add their per-dim log p in LDS (fixed order); wave 0 publishes ONE 32-byte entry per block -- two
//     16-byte stores {l0, l1, tag, 0 | l2, l3, tag, 0}, tag = the group's sequence number.  A half is valid exactly when its
//     own tag says so: no ordering between stores is needed (16-byte stores are observed untorn on gfx950);
//   * every wave sweeps the chunk's A / 4 entries with ONE `sc1` load (it bypasses the CU's L1; lanes 0..31 the first halves,
//     lanes 32..63 the second), finds every tag equal to the sequence it waits for (else polls again, bounded), adds the
//     partials in a fixed order (one lane swap + four DPP steps), runs the PPO head on the four row totals (once: lane l works
//     on row l >> 4) and finishes the backward of THAT group from the e / e*phi it kept in registers;
//   * two groups are in flight per wave: forward of group k, then the backward of group k - 1 with the poll issued at the end
//     of iteration k - 1.  Four slots per (chunk, block), group g in slot g % 4: a wave that writes group k has passed its wait
//     for group k - 1, so every wave of the chunk has published k - 1, i.e. finished reading k - 2 -- and k - 4.
// Where the exchange lives decides everything (measured, same kernel): write-through stores + memory-side loads -- correct
// under any placement -- 704-850 us (every wave of a chunk re-reads the chunk's partials from the memory side once per group:
// 1 GiB of loads on a few hot lines; a tagged hand-off costs ~3 us under streaming load, MI355X_MICROARCH.md row
// handoff-1to1); the same with atomics on one arrival counter per chunk: 6.8 ms (128 arrivals serialise).  Plain stores + `sc1`
// loads keep the exchange INSIDE the chunk's XCD (the stores stay in that XCD's L2, which serves the loads): 448 us.  That
// needs all waves of a chunk on one XCD.  Blocks are dealt to the XCDs round-robin (observed, not promised), so a chunk takes
// the blocks of one blockIdx % 8 class -- and the kernel CHECKS: before the first group every block publishes its XCC id the
// safe way (`sc1` store, memory-side); a chunk whose blocks do not all report the same id runs the safe exchange (slow,
// correct) instead.
// What bounds it after that is the wave's own instruction stream at two waves per SIMD, and in it the VECTOR-MEMORY
// instructions count several times over (measured with alternating builds on one box, profiles/r05_wide_2l_probe.txt):
//   448 -> 415 us   the per-row tails (v_rcp, v_log, the PPO head with its expf) once per group on lane-distributed rows;
//   415 -> 392 us   a chunk takes CONSECUTIVE rows and addresses them through ONE buffer window per wave (a
//                   row that does not exist loads zeros and stores nothing: no clamps, no per-row address arithmetic --
//                   ~250 -> ~100 SALU instructions per group), x and the head's four row scalars arrive in one
//                   lane-distributed load, the exchange is polled by one load instead of two, and nothing is issued that is
//                   then dropped.  On the way: nine more (mostly out-of-range, i.e. dropped) memory instructions per group
//                   cost +6 %, as much as 150 scalar instructions fewer gave back; exact vmcnt waits (every access
//                   unconditional, a true two-iteration prefetch distance) gave nothing by themselves; a second, early poll
//                   in front of the iteration's loads served 45 % of the groups and still lost 3 %; without the lock-step
//                   of the exchange (diagnostic build, wrong results) the kernel is SLOWER (430 us): the waves of a chunk
//                   reading one row together is what the memory system likes; without its dlogits stores it runs 315 us,
//                   the box's copy kernel moves the same bytes in 358 us;
//   392 -> 379 us   the ORDER of the memory instructions (vmcnt retires loads and stores together, in order): a group's row
//                   stores and the prefetch are issued after the exchange's poll, at the top of the next iteration, so that the
//                   wait for the poll is not a wait for them; and the first look at a poll sits outside the retry loop (merged
//                   with the loop's back edge its wait came out as vmcnt(0)).
// The waves of a chunk wait for each other, so the whole grid must be resident at once: the host sizes C to the occupancy
// query and nothing else may occupy the device meanwhile (the package launches on one stream).  Every wait is bounded; a
// wave whose wait expired poisons its gradient slab with NaN (the step's global norm goes NaN: loud, not silent) and sets `err`.
__device__ __forceinline__ void f2l_flag_expiry(unsigned* err, unsigned* sticky) {
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sticky) __hip_atomic_fetch_or(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct WideF2L {
    float* part;        // [C][4 slots][A / 4][8]: the blocks' tagged partial sums (zeroed in front of the launch)
    unsigned* xccs;     // [C][A / 4]: 0x80000000 | XCC id of every block (zeroed in front of the launch)
    unsigned* err;      // [1] this launch's flag (in the ticket area: zeroed in front of every launch)
    unsigned* sticky;   // [1] or nullptr: the CALLER'S fault word -- 1 is OR-ed in, nothing here ever clears it
    float* logp;        // [B] out
    int force_safe;     // tests: 1 = run the memory-side exchange whatever the placement; 2 = spread every chunk over all XCDs
};

template <int M, bool STREAM, bool DBIAS, bool FULLN>
__global__ __launch_bounds__(256, 2) void k_mix_wide_f2l(      // (two waves per SIMD: the whole grid resident at C = 16)
    const float* __restrict__ logits, const float* __restrict__ x,
    const float* __restrict__ loc, const float* __restrict__ logstd,
    int B, int A, int N, int C,
    float* __restrict__ dlogits,
    float* __restrict__ partial,             // [C][PL * A * N] slabs
    float* __restrict__ fin0, float* __restrict__ fin1, float* __restrict__ finb,      // dloc, dlogstd, dbias [A, N]
    WidePpo ppo, WideF2L sy, unsigned* __restrict__ tick)
{
    constexpr int U = 4;
    constexpr int PL = DBIAS ? 3 : 2;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int nb = A >> 2;                                   // blocks per row chunk (A % 4 == 0, nb <= 32)
    // chunk <-> XCD: the blocks of one chunk all carry the same blockIdx % 8 (C % 8 == 0: the host's choice)
    // (force_safe == 2, tests: chunks of CONSECUTIVE blocks instead -- every chunk then straddles all eight XCDs, the placement
    // check below must find that out by itself and the safe exchange must carry the launch)
    const int bi = blockIdx.x >> 3;
    // (force_safe == 3, tests, with an oversubscribed grid: a chunk takes every C-th block, so that its partners span the whole
    // grid and half of them cannot be resident -- the bounded waits must expire.  The shipped mapping keeps a chunk inside 8 nb
    // consecutive blocks: in dispatch order a chunk is complete long before the device is full)
    const int c = __builtin_amdgcn_readfirstlane(sy.force_safe == 2 ? (int)blockIdx.x / nb
                                                 : sy.force_safe == 3 ? (int)blockIdx.x % C : (int)(blockIdx.x & 7) + 8 * (bi / nb));
    const int jb = sy.force_safe == 2 ? (int)blockIdx.x % nb : sy.force_safe == 3 ? (int)blockIdx.x / C : bi % nb;
    const int a = __builtin_amdgcn_readfirstlane(4 * jb + wv);
    const int AN = A * N;
    // FULLN: N == 256 M, every lane owns four particles per float4 slot; else (64 <= N < 256, N % 4 == 0) the lanes past
    // N / 4 idle: their loads are clamped, their exponentials forced to 0, nothing of theirs is stored or summed
    const int nq = N >> 2;
    bool ok[M];
#pragma unroll
    for (int j = 0; j < M; ++j) ok[j] = FULLN || lane + 64 * j < nq;
    const bool has0 = (lane & 31) < nb;                      // lane l sweeps half (l >> 5) of block entry l & 31 of its chunk
    const int d0 = min(lane & 31, nb - 1);
    bool alive = true;
    // ---- placement check: one safe (memory-side) exchange of the blocks' XCC ids ----
    const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(6164) & 15u;           // HW_REG_XCC_ID[3:0]
    bool local = true;
    {
        unsigned* mine = sy.xccs + (size_t)c * nb;
        if (wv == 0 && lane == 0) __hip_atomic_store(mine + jb, 0x80000000u | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int it = 0;
        unsigned seen = 0;
        while (true) {
            seen = __hip_atomic_load(mine + d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(!has0 || (seen >> 31)) || !alive) break;
            __builtin_amdgcn_s_sleep(2);
            if (++it > (1 << 18)) { alive = false; f2l_flag_expiry(sy.err, sy.sticky); }
        }
        local = __all(!has0 || ((seen & 15u) == xcc)) && sy.force_safe != 1;
    }
    v2f b_loc2[M][2], b_isc2[M][2], b_ct2[M][2];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int o = a * N + 4 * (ok[j] ? lane + 64 * j : 0);
        const float4 lo = ld4(loc + o), ls = ld4(logstd + o);
        const float lo_[4] = {lo.x, lo.y, lo.z, lo.w}, ls_[4] = {ls.x, ls.y, ls.z, ls.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = expf(ls_[k]);
            b_loc2[j][k >> 1][k & 1] = lo_[k];
            b_isc2[j][k >> 1][k & 1] = (1.0f / sc) * PFPN_SQRT_HALF_LOG2E;
            b_ct2[j][k >> 1][k & 1] = 64.f - (PFPN_HALF_LOG_2PI + logf(sc)) * PFPN_LOG2E;
        }
    }
    v2f acc_loc2[M][2], acc_ls2[M][2], acc_b2[M][2];
#pragma unroll
    for (int j = 0; j < M; ++j)
        for (int h = 0; h < 2; ++h) {
            acc_loc2[j][h] = (v2f){0.f, 0.f}; acc_ls2[j][h] = (v2f){0.f, 0.f}; acc_b2[j][h] = (v2f){0.f, 0.f};
        }
    const float two_m64 = 5.421010862427522e-20f;
    const float hm = ppo.stats[0], hr = ppo.stats[1], hlo = ppo.stats[2], hhi = ppo.stats[3];
    // A chunk takes CONSECUTIVE rows (the wide kernels deal rows round-robin): the chunk's rows are ONE window per wave -- a
    // buffer descriptor over this wave's action-dim segment in the chunk's rows, row r at byte offset r * pitch in the VGPR
    // offset (range-checked: a row that does not exist -- r < 0 in the first iteration, r >= Rc past the end -- loads zeros
    // and stores nothing, with no clamp, no predicate and no address arithmetic on the scalar unit)
    const int Rq = (B + C - 1) / C;
    const int r0 = c * Rq;
    const int Rc = max(0, min(B, r0 + Rq) - r0);             // rows of chunk c: r0 ... r0 + Rc - 1
    const int G = (Rc + U - 1) / U;                          // its groups of four rows
    const unsigned pitch = (unsigned)AN * 4u;
    const unsigned span = Rc > 0 ? (unsigned)(Rc - 1) * pitch + (unsigned)N * 4u : 0u;    // (< 2^31: the host checks)
    const __amdgpu_buffer_rsrc_t r_lg = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(logits) + (size_t)r0 * AN + (size_t)a * N, 0, (int)span, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dl = __builtin_amdgcn_make_buffer_rsrc(dlogits + (size_t)r0 * AN + (size_t)a * N, 0, (int)span, 0x00020000);
    constexpr int OOB = 0x40000000;                          // an offset beyond any window: the hardware drops the access
    // a lane's offset in row r of the window; a lane past N / 4 (partial grids only) takes an offset no row offset can wrap back
    // into the window (r < 0 gives 2^32 - k * pitch: far out of range by itself, but NOT after adding a constant like OOB)
    auto voff = [&](int r, int j) {
        const unsigned o = (unsigned)r * pitch + 16u * (unsigned)(lane + 64 * j);
        return (FULLN || ok[j]) ? o : 0xFFFFFFF0u;
    };
    float ps = 0.f, vs = 0.f;                                // loss sums of this chunk's rows (kept by its wave a == 0)

    // what a group's backward needs from its forward (z is recomputed from x and the bank: two packed ops per pair)
    // p, 1 / sum e and the head's row scalars stay DISTRIBUTED: lane l holds row l >> 4 of the group's four (what
    // wave_reduce4_dist leaves), so the per-row tails -- v_rcp, v_log, the whole PPO head with its expf -- run ONCE across the
    // four rows instead of four times on wave-uniform values; the backward reads the few scalars it needs with v_readlane
    struct St { v2f e[U][M][2], ef[U][M][2]; float x[U]; float p, inv_sum; float head; };
    // a group's inputs: the logits quads, and ONE lane-distributed load of its rows' scalars: lane l holds input l & 15 of row
    // l >> 4 -- 0..3 the PPO head's (adv, logp_old, value, value_old: a quad-permute broadcast hands the FIRST quad of each
    // row of 16 lanes its row's four; the head's results are only ever read from lanes 0 / 16 / 32 / 48, the other quads
    // compute on whatever they hold), 4..15 the row's action x[row, a] (read out of lane 16 u + 4 with v_readlane) -- loaded
    // with the prefetch, two iterations before its use
    struct Ld { float4 l[U][M]; float xh; };
    const int lj = lane & 15;
    const float* const xh_src = lj == 0 ? ppo.adv : lj == 1 ? ppo.logp_old : lj == 2 ? ppo.value : lj == 3 ? ppo.value_old : x + a;
    const unsigned xh_stride = lj < 4 ? 1u : (unsigned)A;
    auto load = [&](int g, Ld& d) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const u4v q = __builtin_amdgcn_raw_buffer_load_b128(r_lg, voff(U * g + u, j), 0, STREAM ? 2 : 0);     // (2 = nt)
                const f4v f = __builtin_bit_cast(f4v, q);
                d.l[u][j] = make_float4(f.x, f.y, f.z, f.w);
            }
        }
        const unsigned row = min((unsigned)(r0 + U * g + (lane >> 4)), (unsigned)(B - 1));     // (no such row: any row's)
        d.xh = xh_src[(size_t)row * xh_stride];
    };
    // forward of one group: e, e*phi kept; -> the group's four per-dim log p
    auto fwd = [&](const Ld& d, St& s, float& lpr) {
        float mx[U];
        v2f lp[U][M][2];
        float xs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xs[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d.xh), 16 * u + 4));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mx[u] = -INFINITY;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                lp[u][j][0] = (v2f){d.l[u][j].x, d.l[u][j].y} * PFPN_LOG2E;
                lp[u][j][1] = (v2f){d.l[u][j].z, d.l[u][j].w} * PFPN_LOG2E;
                if (ok[j]) {
                    mx[u] = fmaxf(fmaxf(lp[u][j][0].x, lp[u][j][0].y), mx[u]);
                    mx[u] = fmaxf(fmaxf(lp[u][j][1].x, lp[u][j][1].y), mx[u]);
                }
            }
        }
        wave_reduce4<true>(mx[0], mx[1], mx[2], mx[3], lane);
        float sum[U], num[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v2f s2 = {0.f, 0.f}, n2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < M; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f le = lp[u][j][h] - mx[u];
                    const v2f z = (xs[u] - b_loc2[j][h]) * b_isc2[j][h];
                    const v2f arg = le + __builtin_elementwise_fma(-z, z, b_ct2[j][h]);
                    s.e[u][j][h] = (v2f){__builtin_amdgcn_exp2f(le.x), __builtin_amdgcn_exp2f(le.y)};
                    s.ef[u][j][h] = (v2f){__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                    if (!FULLN && !ok[j]) { s.e[u][j][h] = (v2f){0.f, 0.f}; s.ef[u][j][h] = (v2f){0.f, 0.f}; }
                    if (j == 0 && h == 0) { s2 = s.e[u][j][h]; n2 = s.ef[u][j][h]; }
                    else { s2 += s.e[u][j][h]; n2 += s.ef[u][j][h]; }
                }
            }
            sum[u] = s2.x + s2.y;
            num[u] = n2.x + n2.y;
        }
        const float ks = wave_reduce4_dist<false>(sum[0], sum[1], sum[2], sum[3], lane);
        const float kn = wave_reduce4_dist<false>(num[0], num[1], num[2], num[3], lane);
        s.inv_sum = __builtin_amdgcn_rcpf(ks);
        float p = (kn * two_m64) * s.inv_sum;
        if (p < 1.401298464324817e-45f) p = 0.f;
        s.p = p;
        lpr = __builtin_amdgcn_logf(p) * 0.6931471805599453f;               // (lane l: row l >> 4)
#pragma unroll
        for (int u = 0; u < U; ++u) s.x[u] = xs[u];
        s.head = d.xh;                                       // (a copy of a load two iterations old: no wait of its own)
    };
    // the exchange's stores and first-attempt loads are raw BUFFER accesses with the cache policy as an argument (16 = sc1):
    // visible to hipcc's waitcnt pass (an asm load is not).  (vmcnt counts loads and stores together, in order, and wherever
    // paths with different numbers of accesses merge hipcc must assume the fewest: with the exchange's stores behind a
    // wave-uniform branch the loop's waits come out as vmcnt(0) at the poll.  Version 14 issued EVERY access unconditionally --
    // out-of-range offsets for the ones that must not happen -- and got exact waits, vmcnt(24+) for a group's logits: a true
    // two-iteration prefetch distance.  It was no faster, and the dropped accesses cost 6 %: see the header.)
    auto slot_rsrc = [&](int g) {
        // (the base through v_readfirstlane: a descriptor hipcc cannot prove wave-uniform costs a waterfall loop per access)
        const uintptr_t q = (uintptr_t)(sy.part + ((size_t)c * 4 + (g & 3)) * nb * 8);
        const uintptr_t qu = ((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(q >> 32)) << 32) |
                             (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)q);
        return __builtin_amdgcn_make_buffer_rsrc((float*)qu, 0, nb * 32, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t r_logp = __builtin_amdgcn_make_buffer_rsrc(sy.logp, 0, B * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dlogp = __builtin_amdgcn_make_buffer_rsrc(ppo.dlogp, 0, B * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_dvalue = __builtin_amdgcn_make_buffer_rsrc(ppo.dvalue, 0, B * 4, 0x00020000);
    __shared__ float s_blk[2][4][U];
    auto publish = [&](int g, float lpr, bool real) {
        if ((lane & 15) == 0) s_blk[g & 1][wv][lane >> 4] = lpr;
        __syncthreads();                                     // (one per group; buffer g & 1 is rewritten two barriers later)
        float t[U] = {0.f, 0.f, 0.f, 0.f};
        if (wv == 0) {                                       // (LDS only behind this branch)
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = ((s_blk[g & 1][0][u] + s_blk[g & 1][1][u]) + s_blk[g & 1][2][u]) + s_blk[g & 1][3][u];
        }
        const float tag = __uint_as_float((unsigned)g + 1u);
        const __amdgpu_buffer_rsrc_t r = slot_rsrc(g);
        const u4v h0 = __builtin_bit_cast(u4v, (f4v){t[0], t[1], tag, 0.f}), h1 = __builtin_bit_cast(u4v, (f4v){t[2], t[3], tag, 0.f});
        // (an access that is dropped still costs its issue and its pass through the address unit: only the block's first wave
        // issues the entry's two stores -- a wave-uniform branch)
        if (wv == 0) {
            const int o = real && lane == 0 ? jb * 32 : OOB;
            if (local) {                                     // the line stays in this XCD's L2, which serves the readers
                __builtin_amdgcn_raw_buffer_store_b128(h0, r, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(h1, r, o + 16, 0, 0);
            } else {                                         // write-through: readers on other XCDs find it on the memory side
                __builtin_amdgcn_raw_buffer_store_b128(h0, r, o, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b128(h1, r, o + 16, 0, 16);
            }
        }
    };
    // ONE load per wave sweeps the chunk's entries: lanes 0..31 the first halves {l0, l1, tag}, lanes 32..63 the second
    // halves {l2, l3, tag} (the unused fourth dword is not loaded: a dead destination register is re-used at once, and the
    // write to it waits for the load)
    auto poll_issue = [&](int g, f3v& v) {
#if defined(PFPN_F2L_NOX) && PFPN_F2L_NOX == 1
        v = (f3v){0.f, 0.f, 0.f};
        return;
#endif
        v = __builtin_bit_cast(f3v, __builtin_amdgcn_raw_buffer_load_b96(slot_rsrc(g), d0 * 32 + (lane >> 5) * 16, 0, 16));
    };
    // nb partials of four rows -> lane l: the total of row l >> 4, in a fixed order: the lane swap puts rows 0 / 1 (lanes
    // 0..31's two values) and rows 2 / 3 (lanes 32..63's) into their rows of 16 lanes, four DPP steps finish inside the row
    auto row_totals = [&](const f3v& v) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(has0 ? v.x : 0.f), __float_as_uint(has0 ? v.y : 0.f), false, false);
        float k = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        k = row_step_ror8<false>(k);
        k = row_step_ror4<false>(k);
        k = row_step_xor2<false>(k);
        return row_step_xor1<false>(k);
    };
    // the group's four row totals from a poll's values: every half entry must carry the group's tag (else: poll again, bounded)
    auto totals = [&](int g, bool need, const f3v& v, float& tot) {       // -> lane l: the total of row l >> 4
        const unsigned want = (unsigned)g + 1u;
        tot = 0.f;
        if (!need) return;
        int it = 0;
#if defined(PFPN_F2L_NOX)                                    // DIAGNOSTIC build knob (wrong results): nobody waits for the chunk --
        alive = alive && want == 0u;                         // what the lock-step of a chunk's 128 waves costs (1: no poll either)
#endif
        // the first look is at the iteration's own poll, outside the retry loop: its wait is then counted (the accesses issued
        // since -- this iteration's row stores and prefetch -- stay in flight); merged with the loop's back edge, where the poll
        // just issued is the youngest access of the wave, it came out as vmcnt(0)
        if (__all(!has0 || __float_as_uint(v.z) == want)) {
            tot = row_totals(v);
            return;
        }
        f3v w = v;                                           // (retries into a register of their own: `v` is written by the
        while (alive) {                                      //  iteration's poll and by nothing else -- no copies of a poll in flight)
            __builtin_amdgcn_s_sleep(1);
            if (++it > (1 << 18)) {                          // the grid was not co-resident (or a wave died): flagged, poisoned
                alive = false;
                f2l_flag_expiry(sy.err, sy.sticky);
            }
            poll_issue(g, w);
            if (__all(!has0 || __float_as_uint(w.z) == want)) break;
        }
        tot = row_totals(w);
    };
    // PPO head on a group's four rows (k_ppo_loss, term for term; ONCE, lane l working on row l >> 4) + the backward of the
    // group: dlogits of its four rows into `dd` (stored by bwd_store).  Runs for group -1 (zero state) and for the group past the
    // end (clamped rows) too: no row of theirs is live, so q = 0, nothing accumulates and every store falls out of range
    auto bwd = [&](int g, const St& s, float tot, v2f (&dd)[U][M][2]) {
        const int ur = lane >> 4;
        const int b_r = r0 + U * g + ur;
        const bool live_r = (unsigned)(U * g + ur) < (unsigned)Rc;
        // the row's four head inputs out of the first quad of its 16 lanes (meaningful in lanes 16 u ... 16 u + 3 only)
        const float av = dpp_shift<0x00, 0xf>(s.head, 0.f), h_lpo = dpp_shift<0x55, 0xf>(s.head, 0.f);
        const float h_v = dpp_shift<0xAA, 0xf>(s.head, 0.f), h_vo = dpp_shift<0xFF, 0xf>(s.head, 0.f);
        const float nadv = clip_adv((av - hm) * hr, hlo, hhi);
        const float ratio = expf(tot - h_lpo);
        const float sur = ratio * nadv;
        const float csur = fmaxf(fminf(ratio, ppo.clip_hi), ppo.clip_lo) * nadv;
        const bool pass = (sur <= csur) || (ratio >= ppo.clip_lo && ratio <= ppo.clip_hi);
        const float gdl = pass ? (-ppo.inv_B * nadv) * ratio : 0.f;
        const float dvv = h_v - (av + h_vo);
        if (a == 0 && live_r) {
            ps += fminf(sur, csur);                          // (lane l: the sums of rows l >> 4 of every group; joined at the end)
            vs += dvv * dvv;
        }
        // (an access that is dropped still costs its issue and its pass through the address unit: the three per-row outputs
        // are stored by the chunk's first wave only -- a wave-uniform branch; it makes the waits behind it conservative by three)
        if (a == 0)
        {
            const int o3 = a == 0 && live_r && (lane & 15) == 0 ? 4 * b_r : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tot), r_logp, o3, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gdl), r_dlogp, o3, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(2.f * dvv * ppo.value_coef * ppo.inv_B), r_dvalue, o3, 0, 0);
        }
        const float q_ = live_r ? guard_finite(gdl * __builtin_amdgcn_rcpf(s.p)) : 0.f;
        const float qs_d = q_ * s.inv_sum;
        const float qp_d = qs_d * s.p;
        const float qs64_d = qs_d * two_m64;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float qp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qp_d), 16 * u));
            const float qs64 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qs64_d), 16 * u));
#pragma unroll
            for (int j = 0; j < M; ++j) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const v2f tt = s.ef[u][j][h] * qs64;
                    const v2f z = (s.x[u] - b_loc2[j][h]) * b_isc2[j][h];
                    dd[u][j][h] = __builtin_elementwise_fma(-s.e[u][j][h], (v2f){qp, qp}, tt);
                    if (DBIAS) acc_b2[j][h] += dd[u][j][h];
                    acc_loc2[j][h] = __builtin_elementwise_fma(tt, z, acc_loc2[j][h]);
                    acc_ls2[j][h] = __builtin_elementwise_fma(
                        tt, __builtin_elementwise_fma(z * z, (v2f){PFPN_2LN2, PFPN_2LN2}, (v2f){-1.f, -1.f}), acc_ls2[j][h]);
                }
            }
        }
    };
    auto bwd_store = [&](int g, const v2f (&dd)[U][M][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int r = U * g + u;
#if defined(PFPN_F2L_NOSTORE)                                // DIAGNOSTIC build knob (wrong results): dlogits written for one row in 64
            if (r & 63) r = -1;
#endif
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const f4v v = {dd[u][j][0].x, dd[u][j][0].y, dd[u][j][1].x, dd[u][j][1].y};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r_dl, voff(r, j), 0, STREAM ? 16 : 0);
            }
        }
    };

    // Two groups in flight.  Iteration k: the row stores of group k - 2 (computed at the end of iteration k - 1), the forward of
    // group k, the loads of group k + 2 into the buffer the forward has just consumed (two buffers taking turns by NAME: a copy
    // of a freshly loaded register is a wait for the load), publish k, the totals of group k - 1 from the poll issued at the
    // end of iteration k - 1, the head and the backward math of group k - 1, the poll for group k.  The ORDER of the memory
    // instructions is the point: vmcnt retires loads and stores together, in order, so the wait for a poll is a wait for
    // everything issued before it -- with the row stores and the prefetch issued AFTER the poll (at the top of the next
    // iteration) instead of in front of it, the poll has a whole iteration's worth of older traffic ahead of it and nothing
    // younger: 391 -> 379 us.  (Three groups in flight: 496 us; the logits of a third group ahead: 455 us against 448.)
    // Forward and backward run unconditionally -- a guarded one makes every register of its state a select -- iteration 0 works
    // on a zero state, iteration G (+ 1) on rows that load zeros, neither with a live row.
    Ld La, Lb;
    St S0 = {}, S1 = {};
    f3v pv = {0.f, 0.f, 0.f};
    v2f dd[U][M][2] = {};
    float lpr = 0.f, tot = 0.f;
    load(0, La);
    load(1, Lb);
    // (the prologue's loads are waited for HERE, once: the loop header merges this path with the back edge, and a load still in
    // flight on this path -- with next to nothing behind it -- turns into waits at the top of EVERY iteration, vmcnt(8..1), i.e.
    // for the row stores and the prefetch of the iteration before: 391 -> 402 us)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < M; ++j)
            asm volatile("" ::"v"(La.l[u][j].x), "v"(La.l[u][j].y), "v"(La.l[u][j].z), "v"(La.l[u][j].w), "v"(Lb.l[u][j].x),
                         "v"(Lb.l[u][j].y), "v"(Lb.l[u][j].z), "v"(Lb.l[u][j].w));
    asm volatile("" ::"v"(La.xh), "v"(Lb.xh));
#define PFPN_F2L_ITER(KK, CUR, OLD, LC)                                                                      \
    do {                                                                                                     \
        const int kk_ = (KK);                                                                                \
        bwd_store(kk_ - 2, dd);                              /* (computed at the end of the iteration before) */    \
        fwd(LC, CUR, lpr);                                   /* (past the last group: rows that load zeros, never used) */ \
        load(kk_ + 2, LC);                                   /* (into the buffer the forward has just consumed) */  \
        publish(kk_, lpr, kk_ < G);                                                                          \
        totals(kk_ - 1, kk_ >= 1 && kk_ - 1 < G, pv, tot);                                                   \
        bwd(kk_ - 1, OLD, tot, dd);                                                                          \
        poll_issue(kk_, pv);                                                                                 \
    } while (0)
    for (int k = 0; k < G + 1; k += 2) {
        PFPN_F2L_ITER(k, S0, S1, La);
        PFPN_F2L_ITER(k + 1, S1, S0, Lb);
    }
    bwd_store(((G + 2) & ~1) - 2, dd);                       // the rows of the last iteration's backward
#undef PFPN_F2L_ITER

    // ---- the chunk's loss sums; the last chunk adds the C pairs in chunk order (the tail of the forward's hand-off) ----
    if (a == 0) {                                            // rows l >> 4 = 0, 1, 2, 3 of the chunk's groups, in that order
        const auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        ps = ((rl(ps, 0) + rl(ps, 16)) + rl(ps, 32)) + rl(ps, 48);
        vs = ((rl(vs, 0) + rl(vs, 16)) + rl(vs, 32)) + rl(vs, 48);
    }
    if (a == 0 && lane == 0) {
        __hip_atomic_store(ppo.sums + 2 * c, ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ppo.sums + 2 * c + 1, vs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_fetch_add(&tick[WIDE_TICK_PPO], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(C - 1)) {
            float tp = 0.f, tv = 0.f;
            for (int k = 0; k < C; ++k) {
                tp += __hip_atomic_load(ppo.sums + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tv += __hip_atomic_load(ppo.sums + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float pl = -tp * ppo.inv_B, vl = tv * ppo.inv_B * ppo.value_coef;
            ppo.loss3[0] = pl + vl; ppo.loss3[1] = pl; ppo.loss3[2] = vl;
        }
    }
    // ---- slabs + the last wave of an action dim adds the C slabs (k_mix_wide's backward hand-off, same order) ----
    if (!alive) {                                            // an expired wait: results are garbage -- make that loud
#pragma unroll
        for (int j = 0; j < M; ++j) acc_loc2[j][0] = (v2f){NAN, NAN};
    }
    float* slab = partial + (size_t)c * PL * AN + (size_t)a * N;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (!ok[j]) continue;
        const v2f d0 = acc_loc2[j][0] * (b_isc2[j][0] * PFPN_2LN2), d1 = acc_loc2[j][1] * (b_isc2[j][1] * PFPN_2LN2);
        st4_sc1(slab + 4 * (lane + 64 * j), d0.x, d0.y, d1.x, d1.y);
        st4_sc1(slab + AN + 4 * (lane + 64 * j), acc_ls2[j][0].x, acc_ls2[j][0].y, acc_ls2[j][1].x, acc_ls2[j][1].y);
        if (DBIAS) st4_sc1(slab + 2 * AN + 4 * (lane + 64 * j), acc_b2[j][0].x, acc_b2[j][0].y, acc_b2[j][1].x, acc_b2[j][1].y);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(&tick[WIDE_TICK_BWD + a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tk = __builtin_amdgcn_readfirstlane(tk);
    if (tk != (unsigned)(C - 1)) return;
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
        const float* src = partial + (size_t)pl * AN + (size_t)a * N;
        float* dst = (pl == 0 ? fin0 : pl == 1 ? fin1 : finb) + (size_t)a * N;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (!ok[j]) continue;
            f4v run[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) run[r] = (f4v){0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < C; k0 += 16) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f4v v[8];
                    const int kk = k0 + 8 * h;
                    if (kk >= C) break;
                    ld4x8_sc1(src + (size_t)kk * PL * AN + 4 * (lane + 64 * j), (size_t)PL * AN, min(8, C - kk), v);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (kk + i < C) run[8 * h + i] += v[i];
                }
            }
            f4v sum = run[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) sum += run[r];
            *reinterpret_cast<float4*>(dst + 4 * (lane + 64 * j)) = make_float4(sum.x, sum.y, sum.z, sum.w);
        }
    }
}

// logp[b] = sum_a lpseg[b,a]: one wavefront per row, lanes read the row coalesced, fixed-order
// butterfly (deterministic).  (A serial per-thread loop here costs A dependent-load latencies.)
__global__ __launch_bounds__(256) void k_row_sum(const float* __restrict__ lpseg, int B, int A,
                                                 float* __restrict__ logp) {
    const int lane = threadIdx.x & 63;
    const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= B) return;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += lpseg[(size_t)b * A + a];
    s = wave_sum(s);
    if (lane == 0) logp[b] = s;
}

// Fixed-order reduction of per-block slabs: out[i] = sum_k partial[k*stride + i], i < n.
// With `loss_out`, one extra block (the last) reduces the two loss columns behind the n data columns and
// emits loss_out = {loss, policy_loss, value_loss*coef}.  Grid: ceil(n/64) (+1) blocks of 1024.
__global__ __launch_bounds__(1024) void k_finalize(const float* __restrict__ partial, int nslab, int stride, int n,
                                                   float* __restrict__ out0, float* __restrict__ out1, int n0,
                                                   float* __restrict__ loss_out, float inv_B, float value_coef,
                                                   float* __restrict__ out2 = nullptr) {
    __shared__ float red[16][64];
    __shared__ float lsum[2];
    const int cl = threadIdx.x & 63;
    if (loss_out && blockIdx.x == gridDim.x - 1) {
        const float s = slab_reduce_1024<false>(partial, nslab, (size_t)stride, n + (cl & 1), cl < 2, red);
        if ((threadIdx.x >> 6) == 0 && cl < 2) lsum[cl] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float pl = -lsum[0] * inv_B, vl = lsum[1] * inv_B * value_coef;
            loss_out[0] = pl + vl;
            loss_out[1] = pl;
            loss_out[2] = vl;
        }
        return;
    }
    const int i = blockIdx.x * 64 + cl;
    const float s = slab_reduce_1024<false>(partial, nslab, (size_t)stride, i, i < n, red);
    if ((threadIdx.x >> 6) != 0 || i >= n) return;
    if (i < n0) out0[i] = s;
    else if (i < 2 * n0) out1[i - n0] = s;
    else out2[i - 2 * n0] = s;          // only reached with n = 3*n0
}

// adv_stats = {mean, 1/(sqrt(var)+1e-8)} with population variance (reference
// networks/actor_critic/actor_critic.py:151-156).  One block, fixed order.
__global__ __launch_bounds__(1024) void k_adv_stats(const float* __restrict__ adv, int B, int normalize,
                                                    float* __restrict__ stats) {
    __shared__ float red[16];
    if (!normalize) {
        if (threadIdx.x == 0) { stats[0] = 0.f; stats[1] = 1.f; }
        return;
    }
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) s += adv[i];
    const float mean = block_sum(s, red) / (float)B;
    float v = 0.f;
    for (int i = threadIdx.x; i < B; i += blockDim.x) { const float d = adv[i] - mean; v = fmaf(d, d, v); }
    const float var = block_sum(v, red) / (float)B;
    if (threadIdx.x == 0) { stats[0] = mean; stats[1] = 1.0f / (sqrtf(var) + 1e-8f); }
}

// ---------------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------------
struct TilePlan { int R; int grid; int threads; size_t lds; int P; bool ok; int ntiles; };

// The library's only process-wide memory: per-DEVICE caches of two device properties (the CU count, the one-launch PPO pass's
// occupancy), filled on first use.  A slot holds 0 or the final value and two threads racing to fill it write the same value
// (relaxed atomics); nothing a call computes depends on another call.  Devices past the table are asked every time.
constexpr int DEV_SLOTS = 64;
static int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}
// compute units of the CURRENT device (MI355X: 256; a partitioned device reports its own share)
static int cu_count() {
    static std::atomic<int> cache[DEV_SLOTS];
    const int dev = current_device();
    const bool slot = dev >= 0 && dev < DEV_SLOTS;
    int n = slot ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (n == 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        if (slot) cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static bool tile_has(int N) { return N == 35 || N == 50 || N == 16 || N == 8; }

// tile kernel (k_mix_tile): R = 4 rows per tile, 4 action dims per compute wave
static TilePlan plan_tile(int B, int A, int N, const void* logits, const void* dlogits) {
    TilePlan pl{};
    pl.ok = false;
    if (N > 64 || N < 1 || A < 1 || B < 1) return pl;
    const int AN = A * N;
    if ((AN % 4) || ((uintptr_t)logits % 16) || (dlogits && (uintptr_t)dlogits % 16)) return pl;
    const int NC = (A + 3) / 4;
    if (NC + TK_NL + 1 + TK_NS > 16) return pl;         // compute + loader + row + storer waves in one block
    const int R = TK_R;
    pl.R = R;
    pl.threads = NC * 64;                               // compute waves; the launcher adds the service waves of the mode
    const size_t plane = (size_t)(R * AN + ((R * A + 63) & ~63) + 64) * 4;
    const int pmin = dlogits ? 3 : 2;
    int P = 4;                                          // (2-3 planes and > 4 measured no better: LAB_NOTEBOOK.md)
    size_t small = 0;
    for (;; --P) {
        small = (size_t)(4 * R * A + (12 * R + 3 * P) + 8 + 3 * AN) * 4;
        if (P * plane + small <= TK_MAX_LDS || P <= pmin) break;
    }
    pl.P = P;
    pl.lds = P * plane + small;
    const int ntiles = (B + R - 1) / R;
    pl.ntiles = ntiles;
    const int cap = cu_count();                         // one block per CU
    pl.grid = ntiles < cap ? ntiles : cap;
    pl.ok = pl.lds <= TK_MAX_LDS;
    return pl;
}

template <int MODE>
static int launch_tile(int N, const TilePlan& pl, hipStream_t st, const float* logits, const float* x,
                       const float* loc, const float* logstd, const float* dlogp, int B, int A, int tanh_mode,
                       float* logp, float* pmix, float* dlogits, float* dx, float* partial, PpoArgs ppo) {
    const int threads = pl.threads + 64 * (TK_NL + (MODE != MODE_BWD ? 1 : 0) + (MODE != MODE_FWD ? TK_NS : 0));
#define PFPN_LAUNCH_TILE(NN)                                                                                \
    do {                                                                                                    \
        /* > 64 KB of dynamic LDS needs an opt-in, once per device (the attribute belongs to the device's code object) */ \
        static std::atomic<bool> big_lds_ok[DEV_SLOTS];                                                     \
        const int dev_ = current_device();                                                                  \
        const bool slot_ = dev_ >= 0 && dev_ < DEV_SLOTS;                                                   \
        if (pl.lds > 64 * 1024 && !(slot_ && big_lds_ok[dev_].load(std::memory_order_relaxed))) {           \
            (void)hipFuncSetAttribute((const void*)&k_mix_tile<NN, MODE>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)TK_MAX_LDS);        \
            if (slot_) big_lds_ok[dev_].store(true, std::memory_order_relaxed);                             \
        }                                                                                                   \
        hipLaunchKernelGGL((k_mix_tile<NN, MODE>), dim3(pl.grid), dim3(threads), pl.lds, st, logits, x,      \
                           loc, logstd, dlogp, B, A, tanh_mode, pl.P, pl.ntiles / pl.grid,                  \
                           pl.ntiles % pl.grid, logp, pmix, dlogits, dx, partial, ppo);                     \
    } while (0)
    switch (N) {
        case 35: PFPN_LAUNCH_TILE(35); break;
        case 50: PFPN_LAUNCH_TILE(50); break;
        case 16: PFPN_LAUNCH_TILE(16); break;
        case 8: PFPN_LAUNCH_TILE(8); break;
        default: return -1;
    }
#undef PFPN_LAUNCH_TILE
    return 0;
}

// wide path: float4 per lane, up to 4 float4s per lane (N <= 1024)
static bool wide_ok(int A, int N, const void* logits, const void* dlogits) {
    return N >= 64 && N % 4 == 0 && N <= 1024 && ((uintptr_t)logits % 16 == 0) &&
           (dlogits == nullptr || (uintptr_t)dlogits % 16 == 0);
}
static int wide_c_env() { static const int v = getenv("PFPN_WIDE_C") ? atoi(getenv("PFPN_WIDE_C")) : 0; return v; }
// streaming policy of the backward wide kernel (see ld4_stream): on when dlogits cannot stay in the 256 MiB infinity
// cache for the GEMM that follows.  PFPN_WIDE_STREAM=0/1 forces it for A/B runs.
static bool wide_stream(int B, int A, int N) {
    static const int env = getenv("PFPN_WIDE_STREAM") ? atoi(getenv("PFPN_WIDE_STREAM")) : -1;
    if (env >= 0) return env != 0;
    return (size_t)B * A * N * sizeof(float) > ((size_t)256 << 20);
}
// waves per action dim: a whole number of 4-wave blocks per CU over the chip, at least 4 rows per wave, <= 64 slabs
static int wide_chunks(int B, int A, int N, bool bwd = false) {
    // measured on the 8192 x 128 x 256 stress shape (sweeps of PFPN_WIDE_C, +-3 % run to run): ~20 waves/CU
    // forward (C = 38..48: 199-207 us; 32: 209-213; 24: 245; 64: 232); backward, which also streams 1 KiB stores per
    // segment, ~12 waves/CU with the default cache policy (C = 22..24: 419-433 us; 32: 484; 48: 467) and 8 waves/CU
    // with streaming stores (C = 16: 385-404 us; 14: 417; 18: 460-470; 20: 425-437).  More in flight only lengthens
    // the HBM queues (tools/bw_probe.hip), and power-of-two row strides between a wave's rows alias HBM channels.
    const int per_cu = !bwd ? 20 : wide_stream(B, A, N) ? 8 : 12;
    int C = (cu_count() * per_cu + A - 1) / A;
    if (C > 64) C = 64;
    if (wide_c_env() > 0) C = wide_c_env();
    if (C > (B + 3) / 4) C = (B + 3) / 4;      // at least U = 4 rows per wave
    if (C < 1) C = 1;
    return C;
}

}  // namespace pfpn

using namespace pfpn;

// =======================================================================================
// C ABI (declared in include/pfpn_hip.h)
// =======================================================================================
extern "C" {

// diagnostic: install (or clear with NULL) a device buffer of 16*64 uint64 for the phase stamps of block 0
int pfpn_debug_stamps(void* device_buffer) {
    unsigned long long* p = (unsigned long long*)device_buffer;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? PFPN_OK : PFPN_ERR_LAUNCH;
}

size_t pfpn_mixture_workspace_bytes(int B, int A, int N) {
    // slabs: tile path <= 1024 blocks x (3AN+4); wave/wide path C<=64 chunks x 2AN; + lpseg [B*A]
    const size_t AN = (size_t)A * N;
    const size_t slabs = (N <= 64 ? (size_t)1024 : (size_t)64) * (3 * AN + 4);
    // ... + the one-launch PPO pass's exchange area (pfpn_ppo_wide_fused) and, as the LAST WIDE_TICK_WORDS words, the wide
    // kernels' hand-off tickets (wide_tickets()): adjacent, so that the fused pass zeroes both with ONE memset node
    return (((slabs + (size_t)B * A + 64) * sizeof(float) + 15) & ~(size_t)15) + (WIDE_F2L_SYNC_WORDS + WIDE_TICK_WORDS) * sizeof(unsigned);
}

// the ticket area of a caller's workspace: at a place that depends on (B, A, N) alone -- the end of the size
// pfpn_mixture_workspace_bytes() asks for -- or nullptr when the workspace is smaller than that (no in-launch hand-off then)
static unsigned* wide_tickets(void* workspace, size_t workspace_bytes, int B, int A, int N) {
    const size_t need = pfpn_mixture_workspace_bytes(B, A, N);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 3)) return nullptr;
    return (unsigned*)((char*)workspace + need - WIDE_TICK_WORDS * sizeof(unsigned));
}
// zero `n` tickets from `first` on, stream-ordered in front of the launch that draws them (-DPFPN_WIDE_TRUST_TICKETS, A/B build
// knob only: rely on the last arriver having handed them back zeroed)
static bool wide_zero_tickets(unsigned* tick, int first, int n, hipStream_t st) {
#ifdef PFPN_WIDE_TRUST_TICKETS
    return true;
#else
    return hipMemsetAsync(tick + first, 0, (size_t)n * sizeof(unsigned), st) == hipSuccess;
#endif
}

// forward of every path; `ppo` (wide path with the in-launch hand-off only, else PFPN_ERR_ARG): the PPO head in the same launch
static int logprob_fwd_impl(const float* logits, const float* x, const float* loc, const float* logstd,
                            int B, int A, int N, int tanh_mode, float* logp, float* pmix,
                            void* workspace, size_t workspace_bytes, void* stream, const WidePpo* ppo) {
    PFPN_BEGIN();
    if (!logits || !x || !loc || !logstd || !logp || B < 0 || A < 1 || N < 1) return PFPN_ERR_ARG;
    if (B == 0) return PFPN_OK;
    hipStream_t st = (hipStream_t)stream;
    TilePlan pl = plan_tile(B, A, N, logits, nullptr);
    if (pl.ok && tile_has(N)) {
        if (ppo) return PFPN_ERR_ARG;                      // (the tile path has its own fused PPO kernel)
        PpoArgs none{};
        launch_tile<MODE_FWD>(N, pl, st, logits, x, loc, logstd, nullptr, B, A, tanh_mode, logp, pmix,
                                nullptr, nullptr, nullptr, none);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    if (!workspace || workspace_bytes < (size_t)B * A * sizeof(float)) return PFPN_ERR_WORKSPACE;
    float* lpseg = (float*)workspace;
    if (wide_ok(A, N, logits, nullptr)) {
        const int C = wide_chunks(B, A, N);
        const int blocks = (A * C * 64 + 255) / 256;
        const int M = (N / 4 + 63) / 64;
        // the row sum inside the launch (FOLD, see k_mix_wide) whenever a block is four action dims of ONE row chunk and its
        // rows' log p fit in LDS; otherwise per-segment values + the k_row_sum launch
        const int R4 = ((B + C - 1) / C + 3) & ~3;
        const size_t fold_lds = (size_t)4 * R4 * sizeof(float);
        const size_t part_floats = (size_t)C * (A / 4) * R4;
        unsigned* const tick = wide_tickets(workspace, workspace_bytes, B, A, N);
        bool fold = A % 4 == 0 && C <= 64 && fold_lds <= (size_t)WIDE_FOLD_MAX_LDS && tick != nullptr &&
                    (part_floats + 2 * 64) * sizeof(float) <= (size_t)((char*)tick - (char*)workspace);
#ifdef PFPN_WIDE_NO_FOLD                      // build knob for same-box A/B runs (tools/ab_build.sh): the separate launches
        fold = false;
#endif
        if (ppo && !fold) return PFPN_ERR_ARG;
        WidePpo head{};
        if (ppo) { head = *ppo; head.sums = lpseg + part_floats; }
        if (fold && !wide_zero_tickets(tick, WIDE_TICK_FWD, WIDE_TICK_PPO + 1, st)) return PFPN_ERR_LAUNCH;
        // non-temporal loads only in the streaming regime (logits larger than the 256 MiB infinity cache: 207 -> 195 us on
        // the 1 GiB shape); below it the default policy leaves the logits in cache for the backward pass that follows
        const bool stream = wide_stream(B, A, N);
#define PFPN_WIDE_FWD_LAUNCH(MM, TT, FF, SS, OO)                                                                \
    hipLaunchKernelGGL((k_mix_wide<MM, MODE_FWD, TT, FF, SS, OO>), dim3(blocks), dim3(256), OO ? fold_lds : 0, st, \
                       logits, x, loc, logstd, (const float*)nullptr, B, A, N, C, lpseg, pmix, (float*)nullptr,  \
                       (float*)nullptr, (float*)nullptr, logp, (float*)nullptr, head, tick)
#define PFPN_WIDE_FWD_F(MM, TT, FF)                                                                            \
    do {                                                                                                       \
        if (stream) { if (fold) PFPN_WIDE_FWD_LAUNCH(MM, TT, FF, true, true); else PFPN_WIDE_FWD_LAUNCH(MM, TT, FF, true, false); } \
        else { if (fold) PFPN_WIDE_FWD_LAUNCH(MM, TT, FF, false, true); else PFPN_WIDE_FWD_LAUNCH(MM, TT, FF, false, false); } \
    } while (0)
#define PFPN_WIDE_FWD(MM, TT)                                                                                  \
    do {                                                                                                       \
        if (N == 256 * MM) PFPN_WIDE_FWD_F(MM, TT, true); else PFPN_WIDE_FWD_F(MM, TT, false);                  \
    } while (0)
        if (tanh_mode) { if (M == 1) PFPN_WIDE_FWD(1, true); else if (M == 2) PFPN_WIDE_FWD(2, true); else PFPN_WIDE_FWD(4, true); }
        else { if (M == 1) PFPN_WIDE_FWD(1, false); else if (M == 2) PFPN_WIDE_FWD(2, false); else PFPN_WIDE_FWD(4, false); }
#undef PFPN_WIDE_FWD
#undef PFPN_WIDE_FWD_F
#undef PFPN_WIDE_FWD_LAUNCH
        PFPN_CHECK_LAUNCH();
        if (!fold) {
            hipLaunchKernelGGL(k_row_sum, dim3((B * 64 + 255) / 256), dim3(256), 0, st, lpseg, B, A, logp);
            PFPN_CHECK_LAUNCH();
        }
        return PFPN_OK;
    }
    if (ppo) return PFPN_ERR_ARG;
    int C = (B < 64) ? B : 64;
    while ((size_t)A * C < 2048 && C < B) C = (C * 2 < B) ? C * 2 : B;
    const int waves = A * C;
    const int blocks = (waves * 64 + 255) / 256;
    hipLaunchKernelGGL((k_mix_wave<MODE_FWD>), dim3(blocks), dim3(256), 0, st, logits, x, loc, logstd,
                       (const float*)nullptr, B, A, N, C, tanh_mode, lpseg, pmix, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr);
    PFPN_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_row_sum, dim3((B * 64 + 255) / 256), dim3(256), 0, st, lpseg, B, A, logp);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_logprob_fwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     int B, int A, int N, int tanh_mode, float* logp, float* pmix,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return logprob_fwd_impl(logits, x, loc, logstd, B, A, N, tanh_mode, logp, pmix, workspace, workspace_bytes, stream, nullptr);
}

int pfpn_logprob_fwd_ppo(const float* logits, const float* action, const float* loc, const float* logstd,
                         const float* logp_old, const float* adv, const float* value, const float* value_old,
                         const float* adv_stats, int B, int A, int N, float clip_eps, float value_coef,
                         float* logp, float* dlogp, float* dvalue, float* loss3,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!logp_old || !adv || !value || !value_old || !adv_stats || !dlogp || !dvalue || !loss3 || B < 1) return PFPN_ERR_ARG;
    WidePpo p{};
    p.logp_old = logp_old; p.adv = adv; p.value = value; p.value_old = value_old; p.stats = adv_stats;
    p.clip_lo = 1.0f - clip_eps; p.clip_hi = 1.0f + clip_eps; p.value_coef = value_coef; p.inv_B = 1.0f / (float)B;
    p.dlogp = dlogp; p.dvalue = dvalue; p.loss3 = loss3;
    return logprob_fwd_impl(logits, action, loc, logstd, B, A, N, 0, logp, nullptr, workspace, workspace_bytes, stream, &p);
}

// clipped-PPO forward + head + backward of a 256-particle shape in ONE launch, logits read once (k_mix_wide_f2l).
// PFPN_ERR_ARG where the form does not apply (the caller composes pfpn_logprob_fwd_ppo + pfpn_logprob_bwd(_bias)).
int pfpn_ppo_wide_fused(const float* logits, const float* action, const float* loc, const float* logstd,
                        const float* logp_old, const float* adv, const float* value, const float* value_old,
                        const float* adv_stats, int B, int A, int N, float clip_eps, float value_coef,
                        float* logp, float* dlogits, float* dloc, float* dlogstd, float* dvalue, float* dlogp,
                        float* dbias, float* loss3, int* chunks_out, uint32_t* fault_flag,
                        void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !action || !loc || !logstd || !logp_old || !adv || !value || !value_old || !adv_stats || !logp ||
        !dlogits || !dloc || !dlogstd || !dvalue || !dlogp || !loss3 || B < 1 || A < 4 || N < 1)
        return PFPN_ERR_ARG;
    if (!wide_ok(A, N, logits, dlogits) || N > 256 || (A & 3) || A > 128)
        return PFPN_ERR_ARG;                                 // (one float4 slot per lane: 64 <= N <= 256; <= 32 blocks per chunk)
    const bool fulln = N == 256;
    hipStream_t st = (hipStream_t)stream;
    unsigned* const tick = wide_tickets(workspace, workspace_bytes, B, A, N);
    if (!tick) return PFPN_ERR_WORKSPACE;
    const bool stream_policy = wide_stream(B, A, N);
    const int nb = A / 4;
    // every wave of a chunk waits for the others: the WHOLE grid must be resident at once
    // (the occupancy query costs host time: once per device and variant)
    static std::atomic<int> per_cu_cache[DEV_SLOTS][2][2][2];
    const int dev = current_device();
    std::atomic<int>* const per_cu_slot =
        dev >= 0 && dev < DEV_SLOTS ? &per_cu_cache[dev][stream_policy ? 1 : 0][dbias ? 1 : 0][fulln ? 1 : 0] : nullptr;
    int per_cu = per_cu_slot ? per_cu_slot->load(std::memory_order_relaxed) : 0;
#define PFPN_F2L_DISPATCH2(WHAT, SS, DD) do { if (fulln) WHAT(SS, DD, true); else WHAT(SS, DD, false); } while (0)
#define PFPN_F2L_DISPATCH(WHAT)                                                                              \
    do {                                                                                                     \
        if (stream_policy) { if (dbias) PFPN_F2L_DISPATCH2(WHAT, true, true); else PFPN_F2L_DISPATCH2(WHAT, true, false); } \
        else { if (dbias) PFPN_F2L_DISPATCH2(WHAT, false, true); else PFPN_F2L_DISPATCH2(WHAT, false, false); } \
    } while (0)
#define PFPN_F2L_OCC(SS, DD, FF) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_mix_wide_f2l<1, SS, DD, FF>, 256, 0)
    if (per_cu <= 0) {
        PFPN_F2L_DISPATCH(PFPN_F2L_OCC);
        if (per_cu_slot && per_cu > 0) per_cu_slot->store(per_cu, std::memory_order_relaxed);
    }
#undef PFPN_F2L_OCC
    int C = (per_cu * cu_count()) / nb;
    if (wide_c_env() > 0 && wide_c_env() < C) C = wide_c_env();
    // (tests only: a grid TWICE the device's capacity -- half of every chunk's partners cannot be resident, the bounded waits
    // must expire, poison the gradients and set the flag instead of hanging)
    const char* const os_env = getenv("PFPN_WIDE_F2L_OVERSUBSCRIBE");
    if (os_env && atoi(os_env) == 1) C *= 2;
    if (C > 32) C = 32;
    C &= ~7;                                                 // a chunk = the blocks of ONE blockIdx % 8 class
    while (C > 8 && C > (B + 3) / 4) C -= 8;
    if (C < 8 || C > (B + 3) / 4) return PFPN_ERR_ARG;       // (small batches: the two-launch composition)
    // (a wave addresses its chunk's rows through ONE 32-bit buffer window)
    if ((size_t)((B + C - 1) / C) * (size_t)A * (size_t)N * sizeof(float) >= ((size_t)1 << 31)) return PFPN_ERR_ARG;
    if (chunks_out) *chunks_out = C;
    const int AN = A * N;
    const int PLn = dbias ? 3 : 2;
    float* partial = (float*)workspace;
    const size_t slab_floats = (size_t)C * PLn * AN;
    float* const sync = (float*)tick - WIDE_F2L_SYNC_WORDS;                 // [partials | XCC ids | loss sums] | tickets
    if (slab_floats * sizeof(float) > (size_t)((char*)sync - (char*)workspace)) return PFPN_ERR_WORKSPACE;
    const char* const fs_env = getenv("PFPN_WIDE_F2L_SAFE");            // (tests: read at every call)
    const int force_safe = fs_env ? atoi(fs_env) : 0;
    WideF2L sy{};
    sy.part = sync;
    sy.xccs = (unsigned*)(sync + WIDE_F2L_PART_WORDS);
    sy.err = tick + WIDE_TICK_PPO + 1;
    sy.sticky = (unsigned*)fault_flag;
    sy.logp = logp;
    sy.force_safe = force_safe;
    WidePpo p{};
    p.logp_old = logp_old; p.adv = adv; p.value = value; p.value_old = value_old; p.stats = adv_stats;
    p.clip_lo = 1.0f - clip_eps; p.clip_hi = 1.0f + clip_eps; p.value_coef = value_coef; p.inv_B = 1.0f / (float)B;
    p.dlogp = dlogp; p.dvalue = dvalue; p.loss3 = loss3;
    p.sums = sync + WIDE_F2L_PART_WORDS + WIDE_F2L_XCC_WORDS;
    // ONE memset node: the exchange area (tags 0 = nothing published) and the tickets this launch draws
    if (hipMemsetAsync(sync, 0, (WIDE_F2L_SYNC_WORDS + WIDE_TICK_BWD + (size_t)A) * sizeof(unsigned), st) != hipSuccess)
        return PFPN_ERR_LAUNCH;
    const int blocks = C * nb;
#define PFPN_F2L_LAUNCH(SS, DD, FF)                                                                          \
    hipLaunchKernelGGL((k_mix_wide_f2l<1, SS, DD, FF>), dim3(blocks), dim3(256), 0, st, logits, action, loc, logstd, B, A, N, C, \
                       dlogits, partial, dloc, dlogstd, dbias, p, sy, tick)
    PFPN_F2L_DISPATCH(PFPN_F2L_LAUNCH);
#undef PFPN_F2L_LAUNCH
#undef PFPN_F2L_DISPATCH
#undef PFPN_F2L_DISPATCH2
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

static int logprob_bwd_impl(const float* logits, const float* x, const float* loc, const float* logstd,
                            const float* dlogp, int B, int A, int N, int tanh_mode,
                            float* dlogits, float* dloc, float* dlogstd, float* dx, int finalize,
                            void* workspace, size_t workspace_bytes, void* stream, float* dbias) {
    PFPN_BEGIN();
    if (!logits || !x || !loc || !logstd || !dlogp || !dlogits || (finalize && (!dloc || !dlogstd)) || B < 0 || A < 1 ||
        N < 1 || (!finalize && B == 0))
        return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int AN = A * N;
    if (B == 0) {
        (void)hipMemsetAsync(dloc, 0, AN * sizeof(float), st);
        (void)hipMemsetAsync(dlogstd, 0, AN * sizeof(float), st);
        return PFPN_OK;
    }
    if (!workspace) return PFPN_ERR_WORKSPACE;
    float* partial = (float*)workspace;
    TilePlan pl = plan_tile(B, A, N, logits, dlogits);
    if (dbias && !(wide_ok(A, N, logits, dlogits) && !(pl.ok && tile_has(N)) && finalize && A <= WIDE_FOLD_MAX_A))
        return PFPN_ERR_ARG;                                 // (the bias-gradient plane exists in the wide kernel's hand-off only)
    if (pl.ok && tile_has(N)) {
        if (workspace_bytes < (size_t)pl.grid * (2 * AN + 4) * sizeof(float)) return PFPN_ERR_WORKSPACE;
        PpoArgs none{};
        launch_tile<MODE_BWD>(N, pl, st, logits, x, loc, logstd, dlogp, B, A, tanh_mode, nullptr, nullptr,
                                dlogits, dx, partial, none);
        PFPN_CHECK_LAUNCH();
        if (finalize) hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 63) / 64), dim3(1024), 0, st, partial, pl.grid,
                           2 * AN + 4, 2 * AN, dloc, dlogstd, AN, (float*)nullptr, 0.f, 0.f);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    if (wide_ok(A, N, logits, dlogits)) {
        const int C = wide_chunks(B, A, N, true);
        const bool stream = wide_stream(B, A, N);
        if (workspace_bytes < (size_t)C * (dbias ? 3 : 2) * AN * sizeof(float)) return PFPN_ERR_WORKSPACE;
        const int blocks = (A * C * 64 + 255) / 256;
        const int M = (N / 4 + 63) / 64;
        // finalize: the last wave of an action dim adds the slabs inside the launch (k_mix_wide's hand-off) unless A exceeds
        // the ticket array -- then, as before, the k_finalize launch
        unsigned* const tick = wide_tickets(workspace, workspace_bytes, B, A, N);
        bool fold = finalize && A <= WIDE_FOLD_MAX_A && tick != nullptr;
#ifdef PFPN_WIDE_NO_FOLD
        fold = false;
#endif
        if (dbias && !fold) return tick ? PFPN_ERR_ARG : PFPN_ERR_WORKSPACE;
        if (fold && !wide_zero_tickets(tick, WIDE_TICK_BWD, A, st)) return PFPN_ERR_LAUNCH;
        float* const fold0 = fold ? dloc : nullptr;
        float* const fold1 = fold ? dlogstd : nullptr;
#define PFPN_WIDE_BWD_LAUNCH(MM, TT, FF, SS)                                                                    \
    do {                                                                                                        \
        if (dbias)                                                                                              \
            hipLaunchKernelGGL((k_mix_wide<MM, MODE_BWD, TT, FF, SS, true>), dim3(blocks), dim3(256), 0, st, logits, x, loc, \
                               logstd, dlogp, B, A, N, C, dbias, (float*)nullptr, dlogits, dx, partial, fold0, fold1, \
                               WidePpo{}, tick);                                                                \
        else                                                                                                    \
            hipLaunchKernelGGL((k_mix_wide<MM, MODE_BWD, TT, FF, SS>), dim3(blocks), dim3(256), 0, st, logits, x, loc, \
                               logstd, dlogp, B, A, N, C, (float*)nullptr, (float*)nullptr, dlogits, dx, partial, fold0,  \
                               fold1, WidePpo{}, tick);                                                         \
    } while (0)
#define PFPN_WIDE_BWD(MM, TT)                                                                                  \
    do {                                                                                                       \
        if (N == 256 * MM) { if (stream) PFPN_WIDE_BWD_LAUNCH(MM, TT, true, true); else PFPN_WIDE_BWD_LAUNCH(MM, TT, true, false); } \
        else { if (stream) PFPN_WIDE_BWD_LAUNCH(MM, TT, false, true); else PFPN_WIDE_BWD_LAUNCH(MM, TT, false, false); } \
    } while (0)
        if (tanh_mode) { if (M == 1) PFPN_WIDE_BWD(1, true); else if (M == 2) PFPN_WIDE_BWD(2, true); else PFPN_WIDE_BWD(4, true); }
        else { if (M == 1) PFPN_WIDE_BWD(1, false); else if (M == 2) PFPN_WIDE_BWD(2, false); else PFPN_WIDE_BWD(4, false); }
#undef PFPN_WIDE_BWD_LAUNCH
#undef PFPN_WIDE_BWD
        PFPN_CHECK_LAUNCH();
        if (finalize && !fold) hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 63) / 64), dim3(1024), 0, st, partial, C, 2 * AN, 2 * AN,
                           dloc, dlogstd, AN, (float*)nullptr, 0.f, 0.f);
        PFPN_CHECK_LAUNCH();
        return PFPN_OK;
    }
    int C = (B < 64) ? B : 64;
    if (workspace_bytes < (size_t)C * 2 * AN * sizeof(float)) return PFPN_ERR_WORKSPACE;
    const int waves = A * C;
    const int blocks = (waves * 64 + 255) / 256;
    hipLaunchKernelGGL((k_mix_wave<MODE_BWD>), dim3(blocks), dim3(256), 0, st, logits, x, loc, logstd, dlogp,
                       B, A, N, C, tanh_mode, (float*)nullptr, (float*)nullptr, dlogits, dx, partial);
    PFPN_CHECK_LAUNCH();
    if (finalize) hipLaunchKernelGGL(k_finalize, dim3((2 * AN + 63) / 64), dim3(1024), 0, st, partial, C, 2 * AN, 2 * AN,
                       dloc, dlogstd, AN, (float*)nullptr, 0.f, 0.f);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_logprob_bwd(const float* logits, const float* x, const float* loc, const float* logstd,
                     const float* dlogp, int B, int A, int N, int tanh_mode,
                     float* dlogits, float* dloc, float* dlogstd, float* dx, int finalize,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return logprob_bwd_impl(logits, x, loc, logstd, dlogp, B, A, N, tanh_mode, dlogits, dloc, dlogstd, dx, finalize, workspace,
                            workspace_bytes, stream, nullptr);
}

int pfpn_logprob_bwd_bias(const float* logits, const float* x, const float* loc, const float* logstd,
                          const float* dlogp, int B, int A, int N, int tanh_mode,
                          float* dlogits, float* dloc, float* dlogstd, float* dx, float* dbias,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!dbias || B < 1) return PFPN_ERR_ARG;
    return logprob_bwd_impl(logits, x, loc, logstd, dlogp, B, A, N, tanh_mode, dlogits, dloc, dlogstd, dx, 1, workspace,
                            workspace_bytes, stream, dbias);
}

// Slab geometry of pfpn_logprob_bwd(finalize = 0): `nslab` slabs of `stride` floats at the start of its workspace, each
// [dloc A*N | dlogstd A*N | ...]; the caller sums them in slab order (PFPN_JOB_COLSUM does what k_finalize did).
int pfpn_logprob_bwd_slabs(const float* logits, const float* dlogits, int B, int A, int N, int* nslab, int* stride) {
    if (!nslab || !stride || B < 1 || A < 1 || N < 1) return PFPN_ERR_ARG;
    const int AN = A * N;
    TilePlan pl = plan_tile(B, A, N, logits, dlogits);
    if (pl.ok && tile_has(N)) { *nslab = pl.grid; *stride = 2 * AN + 4; return PFPN_OK; }
    if (wide_ok(A, N, logits, dlogits)) { *nslab = wide_chunks(B, A, N, true); *stride = 2 * AN; return PFPN_OK; }
    *nslab = (B < 64) ? B : 64;
    *stride = 2 * AN;
    return PFPN_OK;
}

int pfpn_adv_stats(const float* adv, int B, int normalize, float* stats, void* stream) {
    PFPN_BEGIN();
    if (!adv || !stats || B < 1) return PFPN_ERR_ARG;
    hipLaunchKernelGGL(k_adv_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, adv, B, normalize, stats);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

int pfpn_ppo_mixture_slabs(int B, int A, int N, int want_dbias, const float* logits, const float* dlogits,
                           int* nslab, int* stride) {
    if (!nslab || !stride || !logits || !dlogits || B < 1 || A < 1 || N < 1) return PFPN_ERR_ARG;
    TilePlan pl = plan_tile(B, A, N, logits, dlogits);
    if (!(pl.ok && tile_has(N))) return PFPN_ERR_ARG;
    *nslab = pl.grid;
    *stride = (want_dbias ? 3 : 2) * A * N + 4;
    return PFPN_OK;
}

int pfpn_ppo_mixture_fwd_bwd(const float* logits, const float* action, const float* loc, const float* logstd,
                             const float* logp_old, const float* adv, const float* value,
                             const float* value_old, const float* adv_stats,
                             int B, int A, int N, float clip_eps, float value_coef,
                             float* logp, float* dlogits, float* dloc, float* dlogstd, float* dvalue,
                             float* dbias, float* loss3, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!logits || !action || !loc || !logstd || !logp_old || !adv || !value || !value_old || !adv_stats ||
        !logp || !dlogits || !dloc || !dlogstd || !dvalue || B < 1 || A < 1 || N < 1)
        return PFPN_ERR_ARG;
    if (!workspace) return PFPN_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int AN = A * N;
    TilePlan pl = plan_tile(B, A, N, logits, dlogits);
    if (!(pl.ok && tile_has(N))) return PFPN_ERR_ARG;   // caller falls back to fwd + ppo_loss + bwd
    const int nq = dbias ? 3 : 2;
    if (workspace_bytes < (size_t)pl.grid * (nq * AN + 4) * sizeof(float)) return PFPN_ERR_WORKSPACE;
    PpoArgs pa{};
    pa.logp_old = logp_old; pa.adv = adv; pa.value = value; pa.value_old = value_old; pa.adv_stats = adv_stats;
    pa.inv_B = 1.0f / (float)B; pa.clip_lo = 1.0f - clip_eps; pa.clip_hi = 1.0f + clip_eps;
    pa.value_coef = value_coef; pa.dvalue = dvalue; pa.ratio_out = nullptr; pa.want_db = dbias ? 1 : 0;
    float* partial = (float*)workspace;
    launch_tile<MODE_PPO>(N, pl, st, logits, action, loc, logstd, nullptr, B, A, 0, logp, nullptr, dlogits,
                            nullptr, partial, pa);
    PFPN_CHECK_LAUNCH();
    if (!loss3) return PFPN_OK;          // diagnostic: slabs left unreduced (times the main kernel alone, bench.py)
    hipLaunchKernelGGL(k_finalize, dim3((nq * AN + 63) / 64 + 1), dim3(1024), 0, st, partial, pl.grid, nq * AN + 4,
                       nq * AN, dloc, dlogstd, AN, loss3, pa.inv_B, value_coef, dbias);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

}  // extern "C"
