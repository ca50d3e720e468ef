// fp32 MFMA GEMM with fused dense-layer epilogues for gfx950 (MI355X).
//
// Replaces fc_layer's MatMul + BiasAdd + Relu6 (reference networks/ops.py:105-117) and their TF gradients
// (MatMul_grad, Relu6Grad, BiasAddGrad) with three kernels built on v_mfma_f32_32x32x2_f32 (exact fp32, the
// reference computes in fp32 -- there is no TF32/bf16 shortcut):
//   NN  Y  = act(X W + b)            X [M,K]  W [K,N]            epilogue: +bias, ReLU6
//   NT  dX = (dY W^T) * relu6'(Hin)  dY [M,Kg] W [N,Kg]          epilogue: ReLU6 mask of the layer input
//   TN  dW = X^T dY, db = 1^T dY     X [Kg,M] dY [Kg,N], split over Kg, fixed-order slab reduction
// so the separate bias/activation/mask/bias-grad passes over the activations (~95 us of a 600 us learner step)
// disappear into the GEMMs.
//
// STATUS (round 1): parity-green (tests/test_gpu_gemm.py) but NOT the default dense path: 88-90 TFLOP/s on the
// 2 x [4096x1024x512] layer against 125-143 for the hipBLASLt kernels TunableOp picks (MT128x64x32 MI16x16x4).
// What the measurements say (tools/gemm_probe3.py, tools/mfma_probe.hip, in-kernel s_memtime stamps):
//   * one wave per SIMD issuing nothing but MFMAs reaches 156 TFLOP/s (99 % of peak) with either fp32 MFMA shape, even
//     on a single dependent accumulator chain: occupancy and the MFMA shape are not the limit, and indeed 16x16x4 and
//     32x32x2 give this kernel the same rate, as does the XCD-aware tile order (the operands fit the L2s either way);
//   * inside this kernel the core clock sits at ~2.0 GHz (2.24-2.38 GHz in the pure MFMA loop), a k-tile of 64 MFMAs
//     (4096 cycles) takes ~5500 cycles -- the chunk that carries the 8 LDS stores +730, the chunk with the 8 global
//     loads +300, pure fragment+MFMA chunks +110 each, barrier +125 -- and the epilogue 9 us (128-B row segments);
//   * a fully hand-interleaved variant of the loop (branch-free clamped loads whose k-tail mask is applied at LDS-store
//     time, every LDS store / global load / fragment read placed in an MFMA shadow, barrier before the last chunk) was
//     built and is bit-exact, but no faster (90-98 us): at 2.0 GHz even a stall-free loop needs ~70 us here, which only
//     ties hipBLASLt (60-64 us) + the streaming epilogue pass (7.5 us) it would replace.
// So the learner keeps PyTorch-ROCm GEMMs + learner_kernels.hip; these kernels stay as the fused-epilogue base.
//
// Tiling: 128x128x32 block tile, 256 threads = 2x2 waves, each wave 64x64 = 2x2 MFMA 32x32x2 tiles (64 accumulator
// VGPRs, 4 independent chains: a chain is revisited every 4 x 64 cycles, far beyond the dependent latency).
// LDS staging keeps EVERY store a direct float4 (no transposes, no swizzle arithmetic) and every fragment address
// a per-thread base + compile-time immediate.  A k-tile of 32 is four chunks of 8; MFMA step q of a chunk multiplies
// k = 8c + 4h + q on lane half h = lane >> 5:
//   * K-contiguous operand (rows of memory = rows of the tile): LDS [128][36]; a lane's four k-values of a chunk are
//     ONE ds_read_b128 per 32-row block (36 = 4 mod 32 floats: the 16 rows of a b128 lane group land on 16 disjoint
//     16-B slots -- conflict-free);
//   * tile-contiguous operand (k indexes memory rows): LDS [32][136]; one ds_read_b32 per step (rows k and k+4 are
//     32 banks apart, so the two lane halves never share a bank).
// Both operands of a product use the same k mapping, so the sum over k is unchanged (any bijection works).
// LDS is double buffered (one barrier per k-tile); the next k-tile is prefetched from HBM/L2 into registers and the
// fragments of chunk c+1 are read while the 16 MFMAs of chunk c issue.  Workgroups are re-dealt so that each XCD works
// a contiguous range of tiles (the column tiles of a row panel share its A operand in one L2).
#include <cstdlib>
#include "pfpn_common.h"

namespace pfpn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GBM = 128, GBN = 128, GBK = 32;
constexpr int LDKC = 36;      // K-contiguous operand: LDS [128][36]  (16 consecutive rows -> 16 disjoint 4-bank groups)
constexpr int LDTC = 136;     // tile-contiguous operand: LDS [32][136] (rows k and k+4 are 32 banks apart)
constexpr int KC_FLOATS = 128 * LDKC, TC_FLOATS = GBK * LDTC;

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_BIAS_RELU6 = 2, EPI_MASK = 3 };

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;          // [N]            (EPI_BIAS*)
    const float* H;             // [M, ldh]       (EPI_MASK: activations of the layer input)
    float* colsum;              // TN: [S][N] partial column sums of B (bias gradient), or null
    int M, N, K;                // C is M x N, reduction length K
    int lda, ldb, ldc, ldh;
    long long sA, sB, sC, sBias, sH;    // batch strides (elements)
    int ksplit;                 // TN: number of K chunks (grid.z = batch * ksplit)
    int kchunk;                 // TN: K elements per chunk (multiple of GBK)
    int xcd_remap;              // re-deal tiles so each XCD works a contiguous tile range
};

// One operand of the product: global -> registers -> LDS -> MFMA fragments.
//   KC = true : memory rows are tile rows, k runs along the row (X in X W, dY and W in dY W^T)
//   KC = false: memory rows are k, the tile dimension runs along the row (W in X W, X and dY in X^T dY)
template <bool KC>
struct Operand {
    unsigned goff[4];      // per-thread element offsets of my 4 float4 loads (relative to the k-step base)
    bool gok[4];           // row / column in range
    int kq;                // KC: my k offset inside the k-step (0,4,..,28) ; TC: my k row (0..7) + 8j
    int sts[4];            // LDS float offsets of my 4 float4 stores
    int frag;              // LDS float offset of my fragment base

    __device__ __forceinline__ void init(int ld, int t0, int n_t, int t, int lane, int wbase) {
        const int fhalf = lane >> 5, fcol = lane & 31;     // 32x32x2: lane half h supplies k = 4h + q of a chunk
        if (KC) {
            kq = (t & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (t >> 3) + 32 * j;
                gok[j] = (t0 + row) < n_t;
                goff[j] = (unsigned)(t0 + row) * (unsigned)ld + (unsigned)kq;
                sts[j] = row * LDKC + kq;
            }
            frag = (wbase + fcol) * LDKC + 4 * fhalf;
        } else {
            kq = t >> 5;
            const int cq = (t & 31) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gok[j] = (t0 + cq) < n_t;
                goff[j] = (unsigned)(kq + 8 * j) * (unsigned)ld + (unsigned)(t0 + cq);
                sts[j] = (kq + 8 * j) * LDTC + cq;
            }
            frag = (4 * fhalf) * LDTC + wbase + fcol;
        }
    }
    // P already points at the k-step: KC: P + k0 ; TC: P + k0 * ld
    __device__ __forceinline__ void load(float4 (&r)[4], const float* __restrict__ P, int k0, int kend) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool kok = KC ? (k0 + kq < kend) : (k0 + kq + 8 * j < kend);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gok[j] && kok) v = *reinterpret_cast<const float4*>(P + goff[j]);
            r[j] = v;
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ S, const float4 (&r)[4]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(S + sts[j]) = r[j];
    }
    // fragments of k-chunk c (k = 8c .. 8c+7) for the two 32-wide blocks: f[i][q], q = MFMA k-step inside the chunk.
    // On lane half h the step q multiplies k = 8c + 4h + q: both operands use this mapping, so a chunk still
    // covers its 8 k-values exactly once (the sum over k is order independent up to rounding).
    __device__ __forceinline__ void frags(float (&f)[2][4], const float* __restrict__ S, int c) const {
        if (KC) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(S + frag + 32 * i * LDKC + 8 * c);
                f[i][0] = v.x; f[i][1] = v.y; f[i][2] = v.z; f[i][3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) f[i][q] = S[frag + (8 * c + q) * LDTC + 32 * i];
        }
    }
};

// AKC: A is K-contiguous (row-major [M][K]); else A is stored [K][M].  BKC: B is stored [N][K]; else row-major [K][N].
template <bool AKC, bool BKC, int EPI, bool SPLITK>
__global__ __launch_bounds__(256, 2) void k_gemm(GemmArgs g) {
    constexpr int AF = AKC ? KC_FLOATS : TC_FLOATS;
    constexpr int BF = BKC ? KC_FLOATS : TC_FLOATS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As0 = lds;
    float* Bs0 = lds + AF;
    float* As1 = lds + AF + BF;
    float* Bs1 = lds + 2 * AF + BF;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;       // wave's 64x64 corner in the block tile
    // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by linear id, so neighbouring ids (the
    // column tiles of one row tile, which share the A panel) would land in 8 different L2s.  Re-deal: XCD x works the
    // contiguous id range [x T/8, (x+1) T/8), so a row panel is fetched into one L2 once.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        if (g.xcd_remap && (total & 7u) == 0) {
            const unsigned lid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            unsigned nid = (lid & 7u) * (total >> 3) + (lid >> 3);
            bx = nid % gridDim.x; nid /= gridDim.x;
            by = nid % gridDim.y; bz = nid / gridDim.y;
        }
    }
    const int m0 = by * GBM, n0 = bx * GBN;
    int batch = bz, kc = 0;
    if (SPLITK) { batch = bz / g.ksplit; kc = bz - batch * g.ksplit; }
    const float* A = g.A + (size_t)batch * g.sA;
    const float* B = g.B + (size_t)batch * g.sB;
    const int kbeg = SPLITK ? kc * g.kchunk : 0;
    const int kend = SPLITK ? min(g.K, kbeg + g.kchunk) : g.K;
    const bool do_csum = SPLITK && g.colsum && by == 0;

    Operand<AKC> opA;
    Operand<BKC> opB;
    opA.init(g.lda, m0, g.M, t, lane, wm);
    opB.init(g.ldb, n0, g.N, t, lane, wn);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);              // TN: column sums of my B float4 column group

    float4 ra[4], rb[4];
#define GEMM_FETCH(k_)                                                                          \
    do {                                                                                        \
        opA.load(ra, AKC ? A + (k_) : A + (size_t)(k_) * g.lda, (k_), kend);                   \
        opB.load(rb, BKC ? B + (k_) : B + (size_t)(k_) * g.ldb, (k_), kend);                   \
    } while (0)
#define GEMM_STORE(As_, Bs_)                                                                    \
    do {                                                                                        \
        opA.store((As_), ra);                                                                   \
        opB.store((Bs_), rb);                                                                   \
        if (do_csum) {                                                                          \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                     \
                csum.x += rb[j].x; csum.y += rb[j].y; csum.z += rb[j].z; csum.w += rb[j].w;     \
            }                                                                                   \
        }                                                                                       \
    } while (0)
    const int fhalf = lane >> 5, fcol = lane & 31;
    GEMM_FETCH(kbeg);
    GEMM_STORE(As0, Bs0);
    GEMM_FETCH(kbeg + GBK);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
        const float* Ac = buf ? As1 : As0;
        const float* Bc = buf ? Bs1 : Bs0;
        if (k0 + GBK < kend) {                              // tile k+1 (in registers) -> the other buffer
            if (buf) GEMM_STORE(As0, Bs0); else GEMM_STORE(As1, Bs1);
            GEMM_FETCH(k0 + 2 * GBK);                       // tile k+2 in flight during the MFMAs (past the end: zeros)
        }
        // four k-chunks of 8: the fragments of chunk c+1 are read while the 16 MFMAs (64 cycles each) of chunk c issue
        float fa[2][2][4], fb[2][2][4];
        opA.frags(fa[0], Ac, 0);
        opB.frags(fb[0], Bc, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
                opA.frags(fa[(c + 1) & 1], Ac, c + 1);
                opB.frags(fb[(c + 1) & 1], Bc, c + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c & 1][i][q], fb[c & 1][j][q], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                    // tile k consumed by everyone, tile k+1 fully stored
        buf ^= 1;
    }
#undef GEMM_FETCH
#undef GEMM_STORE
    float* As = As0;

    // ---- epilogue.  32x32 C/D map: col = lane & 31, row = 8 (r >> 2) + 4 (lane >> 5) + (r & 3) -----------------------
    float* C = g.C + (size_t)batch * g.sC + (SPLITK ? (size_t)kc * g.M * g.ldc : 0);   // split-K: slab kc of this batch
    const float* bias = (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU6) ? g.bias + (size_t)batch * g.sBias : nullptr;
    const float* H = (EPI == EPI_MASK) ? g.H + (size_t)batch * g.sH : nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn + 32 * j + fcol;
        if (col >= g.N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + 8 * (r >> 2) + 4 * fhalf + (r & 3);
                if (row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_BIAS_RELU6) v = fminf(fmaxf(v, 0.f), 6.f);
                    if (EPI == EPI_MASK) {
                        const float h = H[(size_t)row * g.ldh + col];
                        v = (h > 0.f && h < 6.f) ? v : 0.f;
                    }
                    C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    }
    if (SPLITK && g.colsum && by == 0) {
        // combine the 8 k-row lanes (t >> 5) that share a column group through LDS, fixed order
        __syncthreads();
        float4* red = reinterpret_cast<float4*>(As);       // 8 x 32 float4 = 4 KiB
        red[(t >> 5) * 32 + (t & 31)] = csum;
        __syncthreads();
        if (t < 32) {
            float4 s = red[t];
            for (int q = 1; q < 8; ++q) { const float4 u = red[q * 32 + t]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
            const int col = n0 + 4 * t;
            if (col < g.N)
                *reinterpret_cast<float4*>(g.colsum + ((size_t)batch * g.ksplit + kc) * g.N + col) = s;
        }
    }
}

// out[b][i] = sum_s slab[b][s][i] in slab order (deterministic); n4 = elements/4 per slab
__global__ __launch_bounds__(256) void k_slab_sum(const float* __restrict__ slab, int S, size_t n, float* __restrict__ out,
                                                  long long out_stride) {
    const size_t n4 = n >> 2;
    const int b = blockIdx.y;
    const float4* src = reinterpret_cast<const float4*>(slab + (size_t)b * S * n);
    float4* dst = reinterpret_cast<float4*>(out + (size_t)b * out_stride);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = src[i];
        for (int k = 1; k < S; ++k) { const float4 u = src[(size_t)k * n4 + i]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
        dst[i] = s;
    }
}

template <bool AKC, bool BKC, int EPI, bool SPLITK>
static void launch_gemm(const GemmArgs& g_in, int batch, hipStream_t st) {
    GemmArgs g = g_in;
    static const int remap = getenv("PFPN_GEMM_XCD_REMAP") ? atoi(getenv("PFPN_GEMM_XCD_REMAP")) : 1;
    g.xcd_remap = remap;
    dim3 grid((g.N + GBN - 1) / GBN, (g.M + GBM - 1) / GBM, batch * (SPLITK ? g.ksplit : 1));
    constexpr size_t lds = (size_t)2 * ((AKC ? KC_FLOATS : TC_FLOATS) + (BKC ? KC_FLOATS : TC_FLOATS)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<AKC, BKC, EPI, SPLITK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gemm<AKC, BKC, EPI, SPLITK>), grid, dim3(256), lds, st, g);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace pfpn

using namespace pfpn;

extern "C" {

// scratch for pfpn_dense_wgrad: split-K slabs of dW [G][S][K][N] and of the column sums [G][S][N]
size_t pfpn_dense_workspace_bytes(int G, int K, int N) {
    return ((size_t)G * 16 * ((size_t)K * N + N) + 64) * sizeof(float);
}

// Y[g] = act(X[g] W[g] + b[g]).  X: [rows, K] (shared by all groups when x_gstride == 0), ldx % 4 == 0, K % 4 == 0;
// W [G, K, N] contiguous, N % 4 == 0; b [G, N]; Y [G, rows, N] contiguous.
int pfpn_dense_fwd(const float* X, long long x_gstride, int ldx, const float* W, const float* bias, int G, int rows,
                   int K, int N, int relu6, float* Y, void* stream) {
    PFPN_BEGIN();
    if (!X || !W || !bias || !Y || G < 1 || rows < 1 || K < 4 || N < 4) return PFPN_ERR_ARG;
    if ((K & 3) || (N & 3) || (ldx & 3) || !aligned16(X) || !aligned16(W) || !aligned16(Y) || (x_gstride & 3))
        return PFPN_ERR_ARG;
    GemmArgs g{};
    g.A = X; g.B = W; g.C = Y; g.bias = bias;
    g.M = rows; g.N = N; g.K = K; g.lda = ldx; g.ldb = N; g.ldc = N;
    g.sA = x_gstride; g.sB = (long long)K * N; g.sC = (long long)rows * N; g.sBias = N;
    if (relu6) launch_gemm<true, false, EPI_BIAS_RELU6, false>(g, G, (hipStream_t)stream);
    else launch_gemm<true, false, EPI_BIAS, false>(g, G, (hipStream_t)stream);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

// dX[g] = (dY[g] W[g]^T) * [0 < Hin[g] < 6] (mask optional: Hin may be NULL).  dY [G, rows, N], W [G, K, N],
// Hin [G, rows, K] (the ReLU6 output that fed this layer), dX [G, rows, K].  K % 4 == 0, N % 4 == 0.
int pfpn_dense_dgrad(const float* dY, const float* W, const float* Hin, int G, int rows, int K, int N, float* dX,
                     void* stream) {
    PFPN_BEGIN();
    if (!dY || !W || !dX || G < 1 || rows < 1 || K < 4 || N < 4) return PFPN_ERR_ARG;
    if ((K & 3) || (N & 3) || !aligned16(dY) || !aligned16(W) || !aligned16(dX)) return PFPN_ERR_ARG;
    GemmArgs g{};
    g.A = dY; g.B = W; g.C = dX; g.H = Hin;
    g.M = rows; g.N = K; g.K = N;                  // output is rows x K, reduction over N
    g.lda = N; g.ldb = N; g.ldc = K; g.ldh = K;
    g.sA = (long long)rows * N; g.sB = (long long)K * N; g.sC = (long long)rows * K; g.sH = (long long)rows * K;
    if (Hin) launch_gemm<true, true, EPI_MASK, false>(g, G, (hipStream_t)stream);
    else launch_gemm<true, true, EPI_NONE, false>(g, G, (hipStream_t)stream);
    PFPN_CHECK_LAUNCH();
    return PFPN_OK;
}

// dW[g] = X[g]^T dY[g]  [K, N]  and  db[g] = column sums of dY[g]  [N].  X [rows, K] (ldx, shared when x_gstride == 0),
// dY [G, rows, N].  The reduction over `rows` is split into chunks whose partial products land in workspace slabs and
// are summed in chunk order (bitwise reproducible).
int pfpn_dense_wgrad(const float* X, long long x_gstride, int ldx, const float* dY, int G, int rows, int K, int N,
                     float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream) {
    PFPN_BEGIN();
    if (!X || !dY || !dW || G < 1 || rows < 1 || K < 4 || N < 4) return PFPN_ERR_ARG;
    if ((K & 3) || (N & 3) || (ldx & 3) || !aligned16(X) || !aligned16(dY) || !aligned16(dW) || (x_gstride & 3))
        return PFPN_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int tiles = ((K + GBM - 1) / GBM) * ((N + GBN - 1) / GBN) * G;
    int S = (1024 + tiles - 1) / tiles;            // ~4 blocks per CU
    const int maxS = (rows + 4 * GBK - 1) / (4 * GBK);
    if (S > maxS) S = maxS;
    if (S > 16) S = 16;
    if (S < 1) S = 1;
    int kchunk = ((rows + S - 1) / S + GBK - 1) / GBK * GBK;
    S = (rows + kchunk - 1) / kchunk;
    const size_t need = ((size_t)G * S * ((size_t)K * N + N)) * sizeof(float);
    if (S > 1 || db) { if (!workspace || workspace_bytes < need) return PFPN_ERR_WORKSPACE; }
    float* slab = (float*)workspace;
    float* cs = slab ? slab + (size_t)G * S * K * N : nullptr;
    GemmArgs g{};
    g.A = X; g.B = dY; g.C = (S > 1) ? slab : dW; g.colsum = db ? cs : nullptr;
    g.M = K; g.N = N; g.K = rows;                  // output K x N, reduction over rows
    g.lda = ldx; g.ldb = N; g.ldc = N;
    g.sA = x_gstride; g.sB = (long long)rows * N; g.sC = (S > 1) ? (long long)S * K * N : (long long)K * N;
    g.ksplit = S; g.kchunk = kchunk;
    launch_gemm<false, false, EPI_NONE, true>(g, G, st);
    PFPN_CHECK_LAUNCH();
    if (S > 1) {
        const size_t n = (size_t)K * N;
        int gx = (int)((n / 4 + 255) / 256);
        if (gx > 1024) gx = 1024;
        hipLaunchKernelGGL(k_slab_sum, dim3(gx, G), dim3(256), 0, st, slab, S, n, dW, (long long)n);
        PFPN_CHECK_LAUNCH();
    }
    if (db) {
        hipLaunchKernelGGL(k_slab_sum, dim3((N / 4 + 255) / 256, G), dim3(256), 0, st, cs, S, (size_t)N, db, (long long)N);
        PFPN_CHECK_LAUNCH();
    }
    return PFPN_OK;
}

}  // extern "C"
