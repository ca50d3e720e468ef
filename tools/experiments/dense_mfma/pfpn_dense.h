/* The C ABI of the two hand-written fp32-MFMA dense families as it stood in include/pfpn_hip.h until round 4
 * (moved out of the product: see README.md in this directory). */
/* ---- dense layers on the fp32 MFMA (v_mfma_f32_32x32x2_f32) with fused epilogues --------------------- */
/* fc_layer forward -- ops.py:105-117: Y[g] = act(X[g] W[g] + b[g]).  X [rows,K] (leading dim ldx, shared by the G
 * groups when x_gstride == 0), W [G,K,N], b [G,N], Y [G,rows,N].  K, N, ldx multiples of 4 (pad with zeros). */
size_t pfpn_dense_workspace_bytes(int G, int K, int N);
int pfpn_dense_fwd(const float* X, long long x_gstride, int ldx, const float* W, const float* bias, int G, int rows,
                   int K, int N, int relu6, float* Y, void* stream);
/* MatMul_grad (input side) fused with Relu6Grad of the layer below: dX[g] = (dY[g] W[g]^T) * [0 < Hin[g] < 6]
 * (Hin NULL: no mask).  dY [G,rows,N], W [G,K,N], Hin/dX [G,rows,K]. */
int pfpn_dense_dgrad(const float* dY, const float* W, const float* Hin, int G, int rows, int K, int N, float* dX,
                     void* stream);
/* MatMul_grad (weight side) fused with BiasAddGrad: dW[g] = X[g]^T dY[g] [K,N], db[g] = column sums of dY[g] [N]
 * (db may be NULL).  Split over `rows` with fixed-order slab reduction (bitwise reproducible). */
int pfpn_dense_wgrad(const float* X, long long x_gstride, int ldx, const float* dY, int G, int rows, int K, int N,
                     float* dW, float* db, void* workspace, size_t workspace_bytes, void* stream);

/* ---- small-row dense layers: ONE launch per layer and direction (<= ~1024 rows) ------------------------------------
 * fc_layer (reference networks/ops.py:82-118: MatMul + BiasAdd + Relu6) and its tf.gradients expansion (MatMul_grad on
 * both sides, Relu6Grad, BiasAddGrad) as a JOB TABLE on the fp32 MFMA: a launch runs up to PFPN_DENSE_MAX_JOBS jobs, a
 * block picks its job from its index, so the weight- and input-gradient products of a layer share one launch.
 * Product jobs, C[g] (M x N, leading dim ldc) for g < batch (operand of group g at base + g * stride):
 *   PFPN_DENSE_NN  C = A B      A [M,K] (lda)  B [K,N] (ldb)      flags BIAS (+ bias[g*sBias + n]), RELU6, MASK,
 *                                                                  HEADPART (aux[row][n / 32] = sum over that 32-column
 *                                                                  block of C[row][n] * hw[n] on group hgroup; hgroup < 0:
 *                                                                  every group, hw + g*sHw, aux + g*sAux)
 *   PFPN_DENSE_NT  C = A Bt^T   A [M,K] (lda)  Bt [N,K] (ldb)     flags MASK: C *= [0 < H[row][n] < 6] (H: ldh, sH)
 *   PFPN_DENSE_TN  C = At^T B   At [K,M] (lda) B [K,N] (ldb)      flags COLSUM: aux[g*sAux + n] = sum_k B[k][n]
 * B, ldb, sB (and N for row-major B, K for Bt) must be 16-byte / 4-float aligned; A may be unaligned (NN, TN: the
 * 197-wide state).  cfg: tile arrangement (0: 64x64, 1: 32x64 with K over 2 waves, 2: 32x32 with K over 4 waves; -1:
 * chosen from the launch's total work).  Sums over K run in a fixed order: bitwise reproducible.
 * Helper jobs on [M]-vectors (a scalar head on top of a layer: critic fc3, actor_critic.py:206-210; twin Q, sac.py:123-139):
 *   PFPN_DENSE_HEAD_FINAL  C[g][b] = bias[g*sBias] + sum_{t<K} A[g][b][t]           (the HEADPART partials, K = ceil(N/32))
 *   PFPN_DENSE_RANK1       C[g][b][n] = A[g][b] * B[g][n] * [0 < H[g][b][n] < 6]    (head backward into the layer below)
 *   PFPN_DENSE_HEAD_WGRAD  C[g][n] = sum_b A[g][b] * H[g][b][n], aux[g*sAux] = sum_b A[g][b]   (dW, db of the head) */
#define PFPN_DENSE_MAX_JOBS 16
#define PFPN_DENSE_NN 0
#define PFPN_DENSE_NT 1
#define PFPN_DENSE_TN 2
#define PFPN_DENSE_HEAD_FINAL 8
#define PFPN_DENSE_RANK1 9
#define PFPN_DENSE_HEAD_WGRAD 10
/* fixed-order reductions riding in the step's LAST dense launch (what pfpn_reduce_jobs does in a launch of its own):
 *   PFPN_DENSE_SLABSUM   C[i] = sum_{k<K} A[k*lda + i] (+ sum_{k<M} B[k*ldb + i] when B != NULL), i < N
 *   PFPN_DENSE_SQSUM     squares of the finished gradient segment A[0..N) (N % 4 == 0, 16-byte aligned) over K blocks
 *   PFPN_DENSE_PPO_LOSS  the two loss columns of pfpn_ppo_mixture_fwd_bwd's slabs (A, stride lda, K slabs) -> C = loss3;
 *                        p0 = 1/B, p1 = value coefficient
 *   PFPN_DENSE_NORM_EMA  A = float64 moments of pfpn_state_prologue [K chunks][2][N], M = batch rows, bias = step,
 *                        B = mean_in, H = std_in, C = mean_out, aux = std_out (utils.py:60-68) */
#define PFPN_DENSE_SLABSUM 11
#define PFPN_DENSE_SQSUM 12
#define PFPN_DENSE_PPO_LOSS 13
#define PFPN_DENSE_NORM_EMA 14
#define PFPN_DENSE_BIAS 1
#define PFPN_DENSE_RELU6 2
#define PFPN_DENSE_MASK 4
#define PFPN_DENSE_COLSUM 8
#define PFPN_DENSE_HEADPART 16
#define PFPN_DENSE_NORM 32        /* the job's outputs are gradients: their squares count (pfpn_dense_jobs_norm) */
typedef struct PfpnDenseJob {
    int kind, flags, cfg;
    int M, N, K, batch;
    int lda, ldb, ldc, ldh;
    int hgroup;
    long long sA, sB, sC, sH, sBias, sAux, sHw;
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* H;
    float* aux;
    const float* hw;
    float p0, p1;
} PfpnDenseJob;
int pfpn_dense_jobs(const PfpnDenseJob* jobs, int n_jobs, void* stream);
/* The same launch also leaving the partial sums of the global gradient norm (tf.clip_by_global_norm, base_worker.py:97-102)
 * for pfpn_clip_adam_prenorm / pfpn_clip_scale_only: block b writes sq_partial[b] = the squares of what it stored for jobs
 * flagged PFPN_DENSE_NORM (dW tiles and their column sums, SLABSUM / HEAD_WGRAD outputs, SQSUM segments), 0 otherwise;
 * *n_partials = the launch's block count (<= sq_capacity).  beta_pow != NULL: block 0 rotates the Adam powers like
 * pfpn_reduce_jobs_norm. */
int pfpn_dense_jobs_norm(const PfpnDenseJob* jobs, int n_jobs, float* sq_partial, size_t sq_capacity, int* n_partials,
                         float* beta_pow, int n_groups, float beta1, float beta2, void* stream);

