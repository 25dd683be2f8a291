/*
 * p2n.h -- C ABI of the MI355X-native Point2Net hot path (libp2n_hip.so).
 *
 * The reference (ahmad-kad/PointCloudSegmentation) has NO native/FFI layer: its
 * boundary for this path is the Python import `from src.models.point2net import
 * Point2Net` (reference train.py:16, inference.py:12) and `from
 * src.models.losses import InstanceSegLoss` (reference train.py:19).  The entry
 * points below are what a ctypes binding inside those modules calls; each one
 * cites the reference lines whose arithmetic it replaces.  INTEGRATION.md shows
 * the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch CUDA tensor data_ptr)
 *     unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - nothing here allocates: outputs and workspaces come from the caller;
 *   - all calls are asynchronous on `stream`; return value 0 = launched OK,
 *     otherwise one of P2N_E*; p2n_last_error() gives a thread-local message;
 *   - float = IEEE binary32, indices = int32_t, layouts are contiguous
 *     row-major in the order written.
 */
#ifndef P2N_H
#define P2N_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2N_OK 0
#define P2N_EINVAL 1      /* bad shape / argument */
#define P2N_EWORKSPACE 2  /* workspace too small */
#define P2N_EHIP 3        /* HIP runtime error (message has hipGetErrorString) */

int p2n_version(void);
const char *p2n_last_error(void);

/* ---- parameters ---------------------------------------------------------
 * Raw parameter buffer = the reference model's state_dict flattened in layer
 * order (reference point2net.py:9-34, 57-65): for l = 0..8 with (Cin,Cout) =
 * (F,64)(64,64)(64,128)(128,128)(128,256)(256,256)(512,256)(256,128)(128,E):
 * W (Cout,Cin), bias (Cout), and for l < 8 bn.weight, bn.bias, running_mean,
 * running_var (Cout each). */
long p2n_param_count(int feature_dim, int embedding_dim);
/* Size (floats) of the eval-mode packed image: weights re-ordered into MFMA
 * fragment order, BatchNorm(eval) folded into per-channel scale/shift. */
long p2n_packed_param_count(int feature_dim, int embedding_dim);
/* Build the packed image from raw params.  Replaces nothing in the reference
 * (it is the load_state_dict-time preparation of reference inference.py:63-74). */
int p2n_pack_eval_params(const float *raw, int feature_dim, int embedding_dim, float *packed, void *stream);

/* ---- eval forward --------------------------------------------------------
 * Point2Net.forward in eval mode (reference point2net.py:67-87; encoder
 * :36-48): x (B,N,F) -> emb (B,N,E).  Optional outputs (NULL to skip):
 * global_f (B,256) = reference point2net.py:46; f1 (B,64,N), f2 (B,128,N),
 * f3 (B,256,N) = the encoder's returned features (reference point2net.py:41-43).
 * emb may be NULL too (encoder-only call: the head kernel is skipped).
 * Supported: 1 <= F <= 8, 1 <= E <= 256, B,N >= 1 (B*N = 0 is a no-op). */
size_t p2n_forward_eval_workspace_bytes(int B, int N, int feature_dim, int embedding_dim);
int p2n_forward_eval(const float *x, int B, int N, int feature_dim, int embedding_dim,
                     const float *packed, float *emb, float *global_f,
                     float *f1, float *f2, float *f3,
                     void *workspace, size_t workspace_bytes, void *stream);

/* Allocation-free variant of p2n_forward_eval.  NO COUNTERPART IN THE REFERENCE: it exists for callers
 * that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  Same arguments and results (bit-identical) as p2n_forward_eval, but the workspace is
 * a device array built into the library (34 MB per GPU, held until the process ends), so the call is
 * limited to B*ceil(N/128) <= P2N_NOALLOC_MAX_TILES workgroups (32,768 points when N is a multiple of
 * 128) and returns P2N_EINVAL beyond.  The call allocates nothing (no hipMalloc/hipFree).  All calls on one
 * GPU share that array: calls that overlap on one GPU (two streams, two threads) race.
 * The array is part of the library's code object: EVERY GPU of EVERY process that loads libp2n_hip.so holds its
 * 34 MB, whether this entry point is ever called or not.  The scratch used is that of the device x lives on,
 * which must be the current device (P2N_EINVAL otherwise).  The first call on a device looks the array's address
 * up (hipGetSymbolAddress) and must therefore happen OUTSIDE stream capture; later calls only launch kernels. */
#define P2N_NOALLOC_MAX_TILES 256
int p2n_forward_eval_noalloc(const float *x, int B, int N, int feature_dim, int embedding_dim,
                             const float *packed, float *emb, float *global_f,
                             float *f1, float *f2, float *f3, void *stream);

/* ---- eval forward in bf16 (opt-in) -------------------------------------------
 * Same contract as p2n_forward_eval (inputs and outputs stay float32) computed on bf16 MFMA
 * with fp32 accumulation (BASELINE north_star: "bf16/fp32 MFMA GEMMs", 2e-2 relative).  Layers 0-2
 * run in split bf16 (hi + lo) to keep the coordinates' precision; measured deviation from the
 * fp32 path: ~6e-3 of max|emb|.  Its packed image is built by p2n_pack_eval_params_bf16
 * (p2n_packed_param_count_bf16 floats); the workspace size is p2n_forward_eval_workspace_bytes. */
long p2n_packed_param_count_bf16(int feature_dim, int embedding_dim);
int p2n_pack_eval_params_bf16(const float *raw, int feature_dim, int embedding_dim, float *packed, void *stream);
int p2n_forward_eval_bf16(const float *x, int B, int N, int feature_dim, int embedding_dim,
                          const float *packed, float *emb, float *global_f,
                          float *f1, float *f2, float *f3,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- "fp32x3": fp32-accurate eval forward on the bf16 matrix pipe (opt-in) ----------------------------
 * Same computation and results as p2n_forward_eval (reference point2net.py:36-48, 57-65, 67-87) to the SAME
 * tolerance (1e-5 of max|ref| per tensor): inputs, outputs, spill and tables are fp32; every product w*x is
 * formed from exact three-way bf16 splits (w = wh+wm+wl, x = xh+xm+xl) as the six part products of weight
 * >= 2^-16, each exact in fp32, accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (the dropped three are below
 * the rounding of the accumulation).  Six bf16 MFMAs replace eight fp32 ones at 1/2 the cycles each.
 * Own packed image and workspace size (workgroups are 256 points).  Python: model.precision = "fp32x3". */
long p2n_packed_param_count_x3(int feature_dim, int embedding_dim);
int p2n_pack_eval_params_x3(const float *raw, int feature_dim, int embedding_dim, float *packed, void *stream);
size_t p2n_forward_eval_x3_workspace_bytes(int B, int N, int feature_dim, int embedding_dim);
int p2n_forward_eval_x3(const float *x, int B, int N, int feature_dim, int embedding_dim,
                        const float *packed, float *emb, float *global_f,
                        float *f1, float *f2, float *f3,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- training path -------------------------------------------------------
 * Train-mode forward (reference point2net.py:36-48,67-87 with model.train(),
 * reference train.py:109,124): BatchNorm uses batch statistics (biased variance),
 * updates running_mean/var in place with `momentum` (unbiased variance), and the
 * activations needed by the backward are kept in `workspace`.
 * params_host: HOST array of 50 DEVICE pointers in raw-parameter order
 *   (W, bias, [bn.weight, bn.bias, running_mean, running_var]) per layer.
 * p2n_train_backward (reference train.py:131, loss.backward()) consumes the same,
 * untouched workspace plus gemb = dL/d emb (B,N,E) and overwrites the 34 gradient
 * tensors grads_host points at, in the order (W, bias, [bn.weight, bn.bias]) per layer.
 * Weight gradients are summed with fp32 atomics (order not reproducible).
 * Memory contract of grads_host: the call writes ONLY the bytes of the 34 tensors, with one exception it can
 * prove harmless: when, sorted by address, the first tensor is 256-byte aligned and every tensor starts exactly at
 * its predecessor's end rounded up to 256 bytes (one flat bucket, as pointcloudsegmentation_amd/train.py hands
 * out), the alignment gaps BETWEEN the tensors are zeroed too (one fill instead of nine).  Any other layout is
 * filled tensor by tensor and nothing outside the tensors is touched.
 * momentum < 0 selects eval-mode BatchNorm for a differentiable forward (model.eval()
 * with autograd enabled, e.g. fine-tuning with frozen statistics): the layers normalise
 * with running_mean/var, which stay untouched, and the backward on that workspace uses
 * dz = gamma*rstd*dy (convolution biases then keep a non-zero gradient).
 * f1, f2, f3 must be NULL: the training forward keeps only pre-BatchNorm activations and does not produce
 * the encoder's feature tensors (P2N_EINVAL otherwise); global_f is optional. */
size_t p2n_train_workspace_bytes(int B, int N, int feature_dim, int embedding_dim);
int p2n_train_forward(const float *x, int B, int N, int feature_dim, int embedding_dim,
                      const void *const *params_host, float momentum, float *emb, float *global_f,
                      float *f1, float *f2, float *f3,
                      void *workspace, size_t workspace_bytes, void *stream);
int p2n_train_backward(const float *x, const float *gemb, int B, int N, int feature_dim, int embedding_dim,
                       const void *const *params_host, void *const *grads_host,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The same two calls in the opt-in bf16 mode (BASELINE north star: "bf16/fp32 MFMA GEMMs", 2e-2): the operands of every
 * GEMM of reference point2net.py:36-48,57-65 and of its backward are rounded to bf16 in the loaders and multiplied on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; BatchNorm statistics, everything stored and all gradients stay
 * fp32.  Same arguments and workspace; a forward and its backward must use the same precision.  Python:
 * model.precision = "bf16". */
int p2n_train_forward_bf16(const float *x, int B, int N, int feature_dim, int embedding_dim,
                           const void *const *params_host, float momentum, float *emb, float *global_f,
                           float *f1, float *f2, float *f3, void *workspace, size_t workspace_bytes, void *stream);
int p2n_train_backward_bf16(const float *x, const float *gemb, int B, int N, int feature_dim, int embedding_dim,
                            const void *const *params_host, void *const *grads_host,
                            void *workspace, size_t workspace_bytes, void *stream);
/* The same two calls in "fp32x3" mode (see p2n_forward_eval_x3): every GEMM operand of reference point2net.py:36-48,57-65 and
 * of its backward is split EXACTLY into three bf16 parts (weights at pack time, activations and gradients in the
 * operand loaders) and the six part products of weight >= 2^-16 accumulate in fp32 on v_mfma_f32_32x32x16_bf16: the
 * fp32 mode's results to fp32 accuracy -- same reference goldens, same tolerances -- at 6/16 of its matrix-pipe cycles.
 * The first layer (reduction length = feature_dim) stays on the fp32 MFMA.  Same arguments and workspace; a forward
 * and its backward must use the same precision.  Python: model.precision = "fp32x3". */
int p2n_train_forward_x3(const float *x, int B, int N, int feature_dim, int embedding_dim,
                         const void *const *params_host, float momentum, float *emb, float *global_f,
                         float *f1, float *f2, float *f3, void *workspace, size_t workspace_bytes, void *stream);
int p2n_train_backward_x3(const float *x, const float *gemb, int B, int N, int feature_dim, int embedding_dim,
                          const void *const *params_host, void *const *grads_host,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Shared MLP of a hierarchical level in train() mode (no reference counterpart: the north star names
 * set-abstraction / feature-propagation levels, the reference has none; semantics = torch's
 * nn.Sequential(Conv1d(k=1), BatchNorm1d, ReLU, ...) in train() mode, parity unpinned by the reference and
 * tested against that torch composition).  The matrix is channel-major over all P columns of the batch
 * ((cloud, point) or (cloud, centroid, neighbour) flattened): x_cm (dims[0], Ppad), out_cm (dims[nl], Ppad),
 * Ppad = P rounded up to 128; dims[0] <= 1024, dims[1..] <= 256, nl <= 8; last_plain != 0: the last layer is a
 * Conv1d alone.  params_host: HOST array of 6 DEVICE pointers per layer (W, bias, bn.weight, bn.bias,
 * running_mean, running_var; the last four NULL for a plain last layer); running statistics are updated in
 * place (momentum in [0,1], unbiased variance, eps 1e-5).  The workspace keeps the pre-BatchNorm activations and
 * must stay untouched until p2n_mlp_train_backward, which takes dout_cm = dL/d out_cm (OVERWRITTEN) and fills
 * grads_host (HOST array of 4 DEVICE pointers per layer: dW, dbias, dgamma, dbeta; the last two NULL for a plain
 * last layer) and dx_cm = dL/d x_cm (dims[0], Ppad), or skips the input gradient when dx_cm is NULL. * momentum < 0: eval-mode BatchNorm under autograd (running statistics used and left untouched), as in
 * p2n_train_forward. */
size_t p2n_mlp_train_workspace_bytes(long P, int nl, const int *dims, int last_plain);
int p2n_mlp_train_forward(const float *x_cm, long P, int nl, const int *dims, int last_plain,
                          const void *const *params_host, float momentum, float *out_cm,
                          void *workspace, size_t workspace_bytes, void *stream);
int p2n_mlp_train_backward(const float *x_cm, float *dout_cm, long P, int nl, const int *dims, int last_plain,
                           const void *const *params_host, void *const *grads_host, float *dx_cm,
                           void *workspace, size_t workspace_bytes, void *stream);

/* InstanceSegLoss (reference losses.py:332-373) = DiscriminativeLoss pull/push
 * (:375-452) + SmoothnessLoss (:454-489, brute-force kNN k+1 with column 0 dropped),
 * values AND gradient w.r.t. the embeddings in one call.
 *   points (B,N,3), emb (B,N,E); inst (B*N): dense id of each point's (cloud,label)
 *   pair, ids ascending in (cloud,label); cnt (T) points per id; bg (T) 1 where the
 *   label is 0 (background, excluded); cstart (B+1) first id of each cloud.
 *   losses (3, device) = {pull, push, smoothness}; instance = pull + push and
 *   total = instance + 0.1*smoothness are left to the caller (losses.py:365).
 *   demb_pull/push/smooth (B,N,E): all three or none (NULL = values only). */
size_t p2n_instance_seg_loss_workspace_bytes(int B, int N, int embedding_dim, int T, int k);
int p2n_instance_seg_loss(const float *points, const float *emb, const int32_t *inst, const int32_t *cnt,
                          const int32_t *bg, const int32_t *cstart, int B, int N, int embedding_dim, int T,
                          float delta_v, float delta_d, int k, float *losses,
                          float *demb_pull, float *demb_push, float *demb_smooth,
                          void *workspace, size_t workspace_bytes, void *stream);
/* The same with the neighbour search (reference losses.py:475-479) done by the caller: knn_idx (B,N,k+1) as written
 * by p2n_knn_self(points, B, N, k+1).  The search depends on the coordinates only, so a training loop can run it on
 * a second stream beside the model's forward (reference train.py:124-127 runs model, then criterion). */
int p2n_instance_seg_loss_knn(const float *points, const float *emb, const int32_t *inst, const int32_t *cnt,
                              const int32_t *bg, const int32_t *cstart, const int32_t *knn_idx, int B, int N,
                              int embedding_dim, int T, float delta_v, float delta_d, int k, float *losses,
                              float *demb_pull, float *demb_push, float *demb_smooth,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Label densification for p2n_instance_seg_loss (reference losses.py:389-395: torch.unique per cloud): labels (B,N)
 * int64 -> inst (B*N) dense ids ascending in (cloud,label); cnt / bg (capacity B*max_per_cloud each, the first
 * cstart[B] entries written); cstart (B+1); overflow (1 int): set when a cloud holds more than max_per_cloud distinct
 * labels (or the value INT64_MAX) -- the outputs are then unusable and the caller takes another route.  Everything
 * stays on the device; the caller reads cstart[B] (= T) and overflow. */
size_t p2n_dense_instances_workspace_bytes(int B, int max_per_cloud);
int p2n_dense_instances(const long long *labels, int B, int N, int max_per_cloud, int32_t *inst, int32_t *cnt,
                        int32_t *bg, int32_t *cstart, int32_t *overflow,
                        void *workspace, size_t workspace_bytes, void *stream);
/* out = (g3[0]*a + g3[1]*b) + g3[2]*c over n floats (n % 4 == 0, 16-byte aligned), g3 on the device: the loss's
 * backward (reference losses.py:365 under autograd: upstream gradients of pull / push / smoothness times the three
 * embedding gradients p2n_instance_seg_loss stored), one pass. */
int p2n_combine3(const float *g3, const float *a, const float *b, const float *c, long n, float *out, void *stream);

/* ---- point-set operators (named by BASELINE.json north_star; the reference has
 * none of them -- SURVEY.md section 0 -- so oracle/p2n_oracle.c is their spec:
 * fp32 squared distance ((dx*dx+dy*dy)+dz*dz) without fma, lowest index wins
 * ties). ----------------------------------------------------------------- */

/* Farthest-point sampling: xyz (B,N,3) -> idx (B,M); first sample = start_idx.
 * Clouds of 4,096 points and more are sampled by a CLUSTER -- single waves (4,096..160,000 points, up to 128 per
 * cloud, kind 1) or 256/1024-thread workgroups (larger clouds, kind 2) -- whose members keep the cloud in registers and
 * exchange records through memory: one exchange per ROUND, and a round accepts every published best that provably is
 * the sequential algorithm's next sample (indices are bit-identical to the one-at-a-time algorithm).  The members of
 * a cluster must all be resident at once (p2n_fps_workgroups_per_cloud > 1); if another kernel keeps them from it, the
 * bounded wait of an exchange runs out: the kernel then stores 1 into *status (device int32, never cleared by the
 * library, so one word can collect many calls; NULL = word 0 of the workspace, cleared by every call) and fills the
 * rest of the cloud's idx row with start_idx.  p2n_fps_cluster_kind: 0 = one workgroup per cloud (cannot fail),
 * 1 = wave members (a few registers and 1-8 KB of LDS each: they fit beside other kernels), 2 = workgroup members
 * (planned as if they owned every compute unit: run them alone). */
int p2n_fps_cluster_kind(int B, int N);
size_t p2n_fps_workspace_bytes(int B, int N, int M);
int p2n_fps_workgroups_per_cloud(int B, int N);
int p2n_fps(const float *xyz, int B, int N, int M, int start_idx, int32_t *idx, int32_t *status,
            void *workspace, size_t workspace_bytes, void *stream);

/* Ball query: xyz (B,N,3), new_xyz (B,M,3) -> idx (B,M,nsample): first nsample
 * indices (ascending) with d2 < radius^2, padded with the first hit (0 if none). */
int p2n_ball_query(const float *xyz, const float *new_xyz, int B, int N, int M,
                   float radius, int nsample, int32_t *idx, void *stream);

/* Brute-force kNN: query (B,M,3), ref (B,N,3) -> idx (B,M,k), d2 (B,M,k) (d2 may
 * be NULL): the k smallest (d2,index) pairs, ascending.  Exact-arithmetic form
 * of reference losses.py:475-479 (cdist + topk(k+1) whose column 0 the caller
 * drops).  1 <= k <= 64, k <= N. */
int p2n_knn(const float *query, const float *ref, int B, int M, int N, int k,
            int32_t *idx, float *d2, void *stream);
/* The same search when the queries ARE the cloud (reference losses.py:475-479 calls
 * cdist(points, points)): xyz (B,N,3) -> idx (B,N,k), d2 (B,N,k) or NULL, results identical
 * to p2n_knn(xyz, xyz, B, N, N, k, ...).  Clouds of up to 16384 points are searched through a
 * Morton-ordered copy with one bounding box per 64 points held in `workspace`; larger ones
 * take p2n_knn's streaming kernel. */
size_t p2n_knn_self_workspace_bytes(int B, int N);
int p2n_knn_self(const float *xyz, int B, int N, int k, int32_t *idx, float *d2,
                 void *workspace, size_t workspace_bytes, void *stream);

/* Grouping gather: feat (B,C,N), idx (B,M,ns) -> out (B,C,M,ns). */
int p2n_group_gather(const float *feat, const int32_t *idx, int B, int C, int N, int M, int ns,
                     float *out, void *stream);
/* Its adjoint: grad_out (B,C,M,ns) scattered-added into grad_feat (B,C,N)
 * (grad_feat is zeroed by the call). */
int p2n_group_gather_bwd(const float *grad_out, const int32_t *idx, int B, int C, int N, int M, int ns,
                         float *grad_feat, void *stream);

/* Three nearest neighbours + fused inverse-distance weights:
 * unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3), idx (B,n,3), weight (B,n,3)
 * (weight may be NULL); w_j = (1/(d2_j+1e-8)) / sum_j. */
int p2n_three_nn(const float *unknown, const float *known, int B, int n, int m,
                 float *dist2, int32_t *idx, float *weight, void *stream);
/* Feature propagation: feat (B,C,m), idx/weight (B,n,3) -> out (B,C,n). */
int p2n_three_interpolate(const float *feat, const int32_t *idx, const float *weight,
                          int B, int C, int m, int n, float *out, void *stream);
int p2n_three_interpolate_bwd(const float *grad_out, const int32_t *idx, const float *weight,
                              int B, int C, int m, int n, float *grad_feat, void *stream);

/* ---- set-abstraction / feature-propagation building blocks (no reference counterpart:
 * the north star names hierarchical levels, the reference has none; parity unpinned).
 * Shared MLP layer = Conv(k=1) + folded BatchNorm(eval) + optional ReLU over any
 * flattened trailing extent L: in (B,Cin,L), W (Cout,Cin), scale/shift (Cout) ->
 * out (B,Cout,L) = act(scale*(W.in)+shift).  Group max-pool: (B,C,M,ns) -> (B,C,M). */
/* Fused SA grouping: xyz (B,N,3), new_xyz (B,M,3), feat (B,C,N) or NULL with C=0,
 * idx (B,M,ns) -> out (B,3+C,M,ns) = [xyz[idx]-new_xyz ; feat[idx]]. */
int p2n_sa_group(const float *xyz, const float *new_xyz, const float *feat, const int32_t *idx,
                 int B, int C, int N, int M, int ns, float *out, void *stream);
size_t p2n_pointwise_layer_workspace_bytes(int Cin, int Cout);
int p2n_pointwise_layer(const float *in, int B, int Cin, long L, const float *W, const float *scale,
                        const float *shift, int relu, int Cout, float *out,
                        void *workspace, size_t workspace_bytes, void *stream);
int p2n_group_max(const float *in, int B, int C, int M, int ns, float *out, void *stream);
/* The same maximum on the CHANNEL-MAJOR output of p2n_mlp_train_forward, for set-abstraction levels in training (no
 * counterpart in the reference: it has no hierarchical levels; replaces y.view(B,C,M,ns).max(dim=3) and its autograd
 * around the train-mode shared MLP): in (C, ld), group g = columns g*ns .. g*ns+ns-1 (g < G, G*ns <= ld, ld % 4 == 0)
 * -> out (C, G) and arg (C, G) = first position of the maximum; the backward writes din (C, ld) = gout at the recorded
 * position, zero elsewhere (pad columns included), i.e. exactly the d out of p2n_mlp_train_backward. */
int p2n_group_max_cm(const float *in, int C, long ld, long G, int ns, float *out, unsigned char *arg, void *stream);
int p2n_group_max_cm_bwd(const float *gout, const unsigned char *arg, int C, long G, int ns, long ld, float *din, void *stream);

/* ---- fused hierarchical levels (no reference counterpart: the north star names set-abstraction and
 * feature-propagation levels, the reference has none; parity unpinned).  One kernel per level: the grouped
 * tensor, the interpolated features and every intermediate layer output stay in registers.  Features
 * between fused levels are POINT-MAJOR: (B,M,C) rows, not (B,C,M).
 * A level's shared MLP is a chain of nl Conv(k=1)+BatchNorm(eval)+ReLU layers, dims[0..nl] its channel
 * counts as the module sees them (dims[0] = 3+C for set abstraction, C1+C2 for propagation); kind 0 = set
 * abstraction (nl 3), 1 = propagation (nl 2), 2 = propagation with the embedding head on top (nl 4, the
 * last layer a plain Conv).  p2n_level_pack builds the chain's weight image (MFMA fragment order,
 * BatchNorm folded in double) once per weight change: params_host = HOST array of 6*nl DEVICE pointers
 * (W (Cout,Cin), bias, bn.weight, bn.bias, running_mean, running_var; the four BatchNorm pointers NULL for
 * a plain layer); rot = number of leading input columns the kernels consume LAST (3 coordinate columns
 * for kind 0, the C1 skip columns for kinds 1/2).  p2n_level_packed_floats returns the image size, or -1
 * when no fused instantiation fits: supported are 3+C <= 136 with mlp <= (128,128,256); C2 = 256 with
 * C1 <= 128, mlp <= (256,256); C2 = 128 with C1 <= 8, mlp <= (128,128,128,64) -- callers then compose the
 * level from the operators above. */
long p2n_level_packed_floats(int kind, int nl, const int *dims, int rot);
int p2n_level_pack(int kind, int nl, const int *dims, const void *const *params_host, int rot, double eps,
                   float *packed, void *stream);
/* Set-abstraction level: xyz rows of ldx floats (first 3 = coordinates), new_xyz (B,M,3), feat rows of ldf
 * floats (first C = features) or NULL with C = 0, idx (B,M,32) ->
 * out (B,M,mlp[2]) = max over the 32 neighbours of MLP([xyz[idx]-new_xyz ; feat[idx]]).  ns must be 32. */
int p2n_sa_level(const float *xyz, int ldx, const float *new_xyz, const float *feat, int ldf,
                 const int32_t *idx, int B, int C, int N, int M, int ns, const int *mlp,
                 const float *packed, float *out, void *stream);
/* Feature-propagation level: idx3/w3 (B,n,3) from p2n_three_nn, feat2 (B,m,C2) coarse features, skip rows
 * of ld_skip floats (first C1 = the fine points' own features) or NULL with C1 = 0 ->
 * out (B,n,mlp[nl-1]) = MLP([skip ; sum_j w3_j * feat2[idx3_j]]); nl = 2, or 4 with the head on top.
 * A level with few points runs layer by layer with its output channels split over the grid (every CU gets a
 * workgroup); the first layer's output then passes through `workspace` (p2n_fp_level_workspace_bytes; 0 when
 * the level runs as one kernel, NULL is then fine). */
size_t p2n_fp_level_workspace_bytes(int B, int n, int C1, int C2, int nl, const int *mlp);
int p2n_fp_level(const int32_t *idx3, const float *w3, const float *feat2, const float *skip, int ld_skip,
                 int B, int n, int m, int C1, int C2, int nl, const int *mlp, const float *packed,
                 float *out, void *workspace, size_t workspace_bytes, void *stream);
/* new_xyz[b][m] = first three floats of row idx[b][m] of xyz (rows of ldx floats): the centroids FPS picked. */
int p2n_gather_xyz(const float *xyz, int ldx, const int32_t *idx, int B, int N, int M, float *out, void *stream);

/* ---- embedding clustering ---------------------
 * p2n_normalize_rows: out[i] = emb[i] / (||emb[i]||_2 + 1e-8) in float32, summed in numpy's
 * pairwise order = reference inference.py:124 bit for bit.  emb, out (N,D), 1 <= D <= 256.
 * p2n_cluster_embeddings: reference inference.py:106-142 (cluster_embeddings): that
 * normalisation, scikit-learn DBSCAN(eps, min_samples) semantics (neighbourhood includes the
 * point, ||a-b|| <= eps decided in float64; clusters numbered by their lowest core index;
 * border points join the lowest-numbered cluster among their core neighbours) and the
 * reference's relabelling: labels (N) int32, 0 = noise, clusters 1..K; *num_clusters = K
 * (device pointer, may be NULL).  eps is a double because the reference passes a Python float.
 * Internally the rows are worked on in the order of a 32-bit code of their embeddings (block
 * pruning needs rows that are close in the embedding to be close in the index); results do not
 * depend on it: labels, cluster numbers and counts refer to the caller's row indices.
 * Environment: P2N_CL_SORT=0 works in the given order, P2N_CL_REACH=0 prunes by bounding spheres
 * only, P2N_CL_SYM=0 lets the count pass visit every block pair from both sides, P2N_CL_NOTES=0
 * switches the link pass's settled-block notes off (all read per call; for tests and A/B timings).
 * p2n_cluster_counts copies out the neighbourhood sizes the last call left in `workspace`. */
int p2n_normalize_rows(const float *emb, long N, int D, float *out, void *stream);
size_t p2n_cluster_workspace_bytes(long N, int D, int min_samples);
int p2n_cluster_embeddings(const float *emb, long N, int D, double eps, int min_samples,
                           int32_t *labels, int32_t *num_clusters,
                           void *workspace, size_t workspace_bytes, void *stream);
int p2n_cluster_counts(const void *workspace, long N, int D, int min_samples, int32_t *counts, void *stream);
/* Slow-path counters of the last call (device uint64[4]): pairs decided in float64 in the count
 * pass, in the link pass, union attempts, border-neighbour records. */
int p2n_cluster_stats(const void *workspace, long N, int D, int min_samples, uint64_t *stats, void *stream);

/* ---- input normalisation -------------------------
 * reference inference.py:76-104 (preprocess_point_cloud): points (N,3) float64, colors (N,3)
 * float64 or NULL (-> 0.5) -> features (N,6) float32 = [(points - mean) / max|points - mean| ,
 * colors (/255 when colors.max() > 1)], bit-identical to numpy's float64 evaluation (the row
 * sum is replayed in numpy's order).  centroid_scale (device double[4], may be NULL) receives
 * {cx, cy, cz, scale}, the reference's second and third return values.
 * features == NULL: statistics only -- `workspace` (8 doubles) then holds {cx, cy, cz, scale, colour flag, ...},
 * the block p2n_forward_eval_raw takes. */
size_t p2n_preprocess_workspace_bytes(void);
int p2n_preprocess_points(const double *points, const double *colors, long N, float *features,
                          double *centroid_scale, void *workspace, size_t workspace_bytes, void *stream);
/* Eval forward of ONE scene straight from the caller's float64 arrays (reference inference.py:88-103 + :172): the
 * normalisation's affine -- subtract the centroid, divide by the scale, colours / 255 when any exceeds 1, concat,
 * .float() -- is applied in layer 0's loader with numpy's own float64 arithmetic, so no (N,6) feature tensor is
 * written and re-read.  prep_cs = the 8-double workspace a p2n_preprocess_points(features = NULL) call filled earlier
 * on the same stream.  emb (N,E) / global_f (256) are bit-identical to p2n_preprocess_points + p2n_forward_eval
 * (feature_dim 6, B = 1); workspace as p2n_forward_eval_workspace_bytes(1, N, 6, E). */
int p2n_forward_eval_raw(const double *points, const double *colors, int N, int embedding_dim, const float *packed,
                         const double *prep_cs, float *emb, float *global_f,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- measurement aid (no counterpart in the reference) ---------------------
 * While enabled, every kernel launch is bracketed by HIP events on its stream.
 * p2n_profile_collect synchronises and fills, per kernel id (0 encoder, 1 mid,
 * 2 head, 3 fps, 5 knn, 10 cluster prep, 11 cluster count pass, 12 cluster link pass), the summed milliseconds and the number of launches since
 * the last collect.  HOST pointers.  Not thread-safe; off by default. */
int p2n_profile_enable(int on);
/* kernel_id >= 0: only that kernel's launches are bracketed (two event records per step instead of
 * two per kernel: the events themselves cost about 3 us each on the launch stream); -1 = all. */
int p2n_profile_select(int kernel_id);
int p2n_profile_collect(int max_ids, float *ms_sum_host, int *count_host);

#ifdef __cplusplus
}
#endif
#endif /* P2N_H */
