/* re2e.h -- C ABI of libre2e_hip.so: MI355X (gfx950) kernels for the joint_train hot path of
 * bliunlpr/Robust_e2e_gan.
 *
 * The reference has NO native interface on this path (all arithmetic is stock PyTorch-0.4 ATen
 * ops plus warp-ctc, model/e2e_ctc.py:63), so each entry point cites the reference CALL SITE
 * whose ATen/warp-ctc op it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (fp32 unless named *_i32 / *_u8);
 *     the library never allocates, frees or retains device memory; scratch is caller-provided;
 *     `lens_host` arguments (where present) are HOST int32 arrays read during the call only;
 *   - asynchronous: work is enqueued on `stream`; no device synchronisation, no host read-back;
 *   - returns RE2E_OK (0) or a negative RE2E_E* code; never throws/exits; the message for the
 *     last failure on the calling thread is re2e_last_error();
 *   - reductions feeding parity-checked outputs are fixed-tree (bitwise reproducible run to run);
 *   - activations: NHWC for the conv stacks, (T,B,F) time-major inside the recurrent stacks,
 *     (B,T,F) batch-first at the module boundary, exactly as the reference lays them out.
 */
#ifndef RE2E_H
#define RE2E_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* re2e_stream_t; /* == hipStream_t */

#define RE2E_OK 0
#define RE2E_EINVAL (-1)
#define RE2E_EUNSUPPORTED (-2)
#define RE2E_EHIP (-3)

#define RE2E_ACT_NONE 0
#define RE2E_ACT_TANH 1
#define RE2E_ACT_RELU 2
#define RE2E_ACT_LRELU 3 /* slope 0.2 (model/gan_model.py:65) */
#define RE2E_ACT_SIGMOID 4
#define RE2E_ACT_SIGMOID_MASK_MUL 5 /* sigmoid -> length mask -> * mix (model/enhance_model.py:156-164) */

#define RE2E_LOSS_L2 0
#define RE2E_LOSS_L1 1
#define RE2E_LOSS_SMOOTH_L1 2
#define RE2E_STREAM_DEFAULT 0
#define RE2E_STREAM_FILLER 1 /* bulk work overlapped with resident recurrences: see re2e_stream_role */

#define RE2E_LOSS_BCE 3 /* nn.BCELoss on probabilities, logs clamped at -100 (model/gan_model.py:157-160, --no_lsgan) */

/* ABI version of this header: bumped whenever an entry point is added or a signature changes (positional arguments carry no
 * names across the boundary).  re2e_version() returns the value the library was built with; a binding written for another value
 * must refuse to call (robust_e2e_gan_amd/lib.py load()). */
#define RE2E_ABI_VERSION 319
int re2e_version(void);
const char* re2e_last_error(void);
/* 1 when device 0 is gfx950, 0 when another arch, <0 on HIP error. */
int re2e_device_ok(void);
/* Optional, no reference counterpart: prepares every persistent-recurrence kernel (code object loaded, dynamic-LDS limit raised) so
 * that the first sequence of a process is not a ~28 ms launch.  No stream, no device work, idempotent; needs a current HIP device. */
int re2e_warmup(void);
/* Scheduling hint, no reference counterpart (the reference runs one stream).  role = RE2E_STREAM_FILLER marks a stream whose
 * kernels run BESIDE the persistent recurrences of another stream (JointTrainer's side / weight-gradient streams,
 * joint_train.py step): the MFMA engine then launches 4-wave tiles there, which fit the registers a resident recurrence
 * leaves free on a CU; RE2E_STREAM_DEFAULT removes the mark.  Results do not depend on it.  At most 64 filler streams at a time. */
int re2e_stream_role(re2e_stream_t stream, int role);

/* ---- K3 dense GEMM (fp32 MFMA).  C[M,N] = act(op(A) op(B) + bias + bias2) + beta*C.
 * Row-major; transa=0: A stored [M,K], transa=1: A stored [K,M]; transb=0: B stored [K,N],
 * transb=1: B stored [N,K].
 *   (0,1)  x W^T     -- torch.nn.Linear forward  (e2e_encoder.py:145,173)
 *   (0,0)  dy W      -- input gradient
 *   (1,0)  dy^T x    -- weight gradient (split-K over rows, needs workspace)
 *   (1,1)  unsupported (RE2E_EUNSUPPORTED)
 * act=RE2E_ACT_SIGMOID_MASK_MUL writes mask_out=sigmoid(.)*[t<lens[b]] and C=mask_out*mul,
 * rows being (b,t) batch-first with T frames per utterance. */
size_t re2e_gemm_workspace_bytes(int transa, int transb, int M, int N, int K);
int re2e_gemm(int transa, int transb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
              float* C, long ldc, const float* bias, const float* bias2, int act, float beta, const float* mul,
              float* mask_out, const int* lens_dev, int T, void* workspace, size_t workspace_bytes,
              re2e_stream_t stream);
/* x W^T over the VALID rows of a ragged time-major batch (the reference packs its sequences: e2e_encoder.py:129-131, enhance_model.py:120-123):
 *   C[map[r]][:] = act(A[map[r]][:] . B[N,K]^T + bias + bias2) + beta C[map[r]][:]   for r < Mv,
 * A (phys_rows x K, lda) and C (phys_rows x N, ldc) in the padded layout, rowmap[Mv] the physical rows with t < len_b.  Rows outside the map are
 * not touched (re2e_fill_rows).  Workspace as re2e_gemm(0, 1, Mv, N, K).  RE2E_EUNSUPPORTED when the shape is not one the LDS-DMA pipeline
 * takes (K, N, lda, ldb, ldc multiples of 4, 16-byte aligned operands, Mv >= 256): the caller runs re2e_gemm over all rows instead.
 * ident_rows (0 .. Mv, both entry points): the caller's promise that rowmap[r] == r for r < ident_rows (every utterance is at least that long:
 * min(len) * B rows of a time-major batch) -- tiles / k-tiles below it skip the table; 0 is always correct. */
int re2e_gemm_nt_rows(int Mv, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias,
                      const float* bias2, int act, float beta, const int* rowmap, int ident_rows, int phys_rows, void* workspace,
                      size_t workspace_bytes, re2e_stream_t stream);
/* The weight gradient over the same rows: C[M,N] = sum_{r < Kv} A[map[r]][:M]^T B[map[r]][:N] + beta C, A (dy) and B (x) padded
 * (phys_rows x ., lda / ldb).  Workspace as re2e_gemm(1, 0, M, N, Kv).  RE2E_EUNSUPPORTED unless both operands are 16-byte loadable. */
int re2e_gemm_tn_rows(int M, int N, int Kv, const float* A, long lda, const float* B, long ldb, float* C, long ldc, float beta,
                      const int* rowmap, int ident_rows, int phys_rows, void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* C[rows[i]][0 .. N) = row_vec ? act(row_vec[0 .. N)) : value, i < nrows (N, ldc multiples of 4; row_vec 16-byte aligned, act one of
 * RE2E_ACT_NONE .. RE2E_ACT_SIGMOID).  With row_vec = the product's bias the padded rows hold what the reference's Linear + activation over
 * zero-padded frames leaves there: tanh(bias), e2e_encoder.py:145-147,173-176. */
int re2e_fill_rows(float* C, long ldc, int N, const int* rows, int nrows, float value, const float* row_vec, int act, re2e_stream_t stream);

/* ---- K5/K9 convolution as implicit GEMM over NHWC (nn.Conv2d: e2e_encoder.py:234-237,
 * gan_model.py:63-90).  `wg` is the gathered weight [Cout][KH][KW][C] from
 * re2e_conv_weight_gather.  Pixel (n,py,px) of the logical PHxPW grid reads
 * in[n][py*SY+kh*DY+OY0][px*SX+kw*DX+OX0][:] and is written at
 * out[n][py*osy+ooy][px*osx+oox][:] of an (NI,OHF,OWF,Cout) tensor -- forward: S=stride,D=1,
 * O0=-pad; stride-1 data gradient: S=1,D=-1,O0=+pad with transposed weights; stride-2 data
 * gradient: one call per output parity class. */
int re2e_conv_igemm(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, int KH, int KW, int PH,
                    int PW, int SY, int SX, int DY, int DX, int OY0, int OX0, float* out, int OHF, int OWF, int osy,
                    int osx, int ooy, int oox, const float* bias, int act, float beta, re2e_stream_t stream);
/* The same product (no bias, no activation, dense output) followed by out = relu_out > 0 ? out : 0, `relu_out` an (NI,PH,PW,Cout)
 * tensor: the data gradient of a convolution whose INPUT was the ReLU output `relu_out` of the layer in front, taken through that
 * ReLU in the same launch (e2e_encoder.py:259-265: conv1_1 -> ReLU -> conv1_2, conv2_1 -> ReLU -> conv2_2), so the layer in front
 * needs no activation-backward pass.  3x3 / stride-1 geometries apply the mask in the kernel's epilogue, others in a second pass. */
int re2e_conv_igemm_masked(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, int KH, int KW, int PH,
                           int PW, int SY, int SX, int DY, int DX, int OY0, int OX0, float* out, int OHF, int OWF, int osy,
                           int osx, int ooy, int oox, const float* relu_out, re2e_stream_t stream);
/* 3x3 / stride-1 / pad-1 convolution -> ReLU -> 2x2 / stride-2 ceil-mode max pool in ONE launch (e2e_encoder.py:260-262, 264-266): only the
 * pooled tensor (NI, ceil(H/2), ceil(W/2), Cout) and its index bytes (re2e_maxpool2_fwd's, with relu_in = 1) are written, the
 * full-resolution activation never reaches memory.  RE2E_EUNSUPPORTED unless C % 16 == 0, Cout % 64 == 0 and the tensors are 16-byte
 * aligned and < 2 GiB (the caller then runs re2e_conv_igemm + re2e_maxpool2_fwd). */
int re2e_conv3x3_relu_pool(const float* in, int NI, int H, int W, int C, const float* wg, int Cout, const float* bias, float* pooled,
                           unsigned char* idx_u8, re2e_stream_t stream);
/* 3x3 / stride-1 / pad-1 convolution as a fused Winograd F(2x2,3x3) on the fp32 matrix core (2.25x fewer matrix instructions than the
 * direct form; the transforms only add and halve, results agree with re2e_conv_igemm to a few ulp).  `w` is the layer's weight in
 * PyTorch's (Cout_f, Cin_f, 3, 3) layout -- no gathered copy is needed.  dgrad = 0: forward, in (NI,H,W,C = Cin_f) -> out
 * (NI,H,W,Cout = Cout_f), optional bias (Cout) and ReLU (relu = 1).  dgrad = 1: data gradient, in = dy (NI,H,W,C = Cout_f) -> out = dx
 * (NI,H,W,Cout = Cin_f).  `mask` (optional, shape of out): out = mask > 0 ? value : 0 (re2e_conv_igemm_masked's epilogue).
 * `pool_out` / `pool_idx` (optional, forward + relu): ONLY maxpool2(relu(conv)) (NI, ceil(H/2), ceil(W/2), Cout) and its index bytes are
 * written (re2e_conv3x3_relu_pool's contract), `out` may then be NULL.  The workspace holds the transformed weights (prepared by the
 * call itself).  RE2E_EUNSUPPORTED unless C % 8 == 0, Cout % 64 == 0, 16-byte aligned operands and tensors < 2 GiB: the caller then
 * uses re2e_conv_igemm / re2e_conv_igemm_masked / re2e_conv3x3_relu_pool (e2e_encoder.py:234-237,258-266 and their autograd gradients). */
size_t re2e_conv3x3_wino_workspace_bytes(int C, int Cout);
int re2e_conv3x3_wino(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int dgrad, const float* bias, int relu,
                      const float* mask, float* out, float* pool_out, unsigned char* pool_idx, void* workspace, size_t workspace_bytes,
                      re2e_stream_t stream);

/* Weight gradient of the same layers the same way (csrc/wino_wgrad.hip): gw (Cout, C, 3, 3) (+)= G^T [ sum over 2x2 output tiles of
 * (B^T d B)[c] (x) (A dy A^T)[o] ] G -- transform + product in one kernel (split over patch ranges into `workspace` slabs), a second
 * one sums the slabs and applies G^T . G.  in (NI,H,W,C), dout (NI,H,W,Cout) NHWC, beta 0 / 1; C % 64 == 0, Cout % 32 == 0,
 * tensors < 2 GiB and 16-byte aligned, else RE2E_EUNSUPPORTED (the caller uses re2e_conv_wgrad). */
size_t re2e_conv3x3_wino_wgrad_workspace_bytes(int NI, int H, int W, int C, int Cout);
int re2e_conv3x3_wino_wgrad(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, float* gw, float beta,
                            void* workspace, size_t workspace_bytes, re2e_stream_t stream);

/* The three over a RAGGED image batch (the VGG front end convolves zero-padded utterances and cuts each at its pooled length afterwards,
 * e2e_encoder.py:259-278: what it computes further beyond an utterance's end than the stack reaches is never read).  row_lim[n], int32 on
 * the device: output rows y >= row_lim[n] of image n (full-resolution rows, also with the fused pool) are neither computed nor written by
 * re2e_conv3x3_wino_rows -- patches of 8 / 16 rows that START there are skipped, so limits are multiples of 16; the weight gradient skips
 * the patches that start there (its dout is zero there: the caller's promise).  re2e_fill_image_rows writes `value` into rows
 * r >= ceil(lim[n] / div) of a (N, H, row_floats) tensor (div = 2: the pooled output of a layer whose limits count full-resolution rows);
 * max_tail_rows bounds H - lim / div over the batch.  The caller keeps the limits of consecutive layers consistent (VGG2L.conv_stack). */
int re2e_conv3x3_wino_rows(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int dgrad, const float* bias, int relu,
                           const float* mask, float* out, float* pool_out, unsigned char* pool_idx, const int* row_lim, void* workspace,
                           size_t workspace_bytes, re2e_stream_t stream);
int re2e_conv3x3_wino_wgrad_rows(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, float* gw, float beta,
                                 const int* row_lim, void* workspace, size_t workspace_bytes, re2e_stream_t stream);
int re2e_fill_image_rows(float* t, int N, int H, long row_floats, const int* lim, int div, int max_tail_rows, float value, re2e_stream_t stream);

/* 4x4 / stride-1 convolution as Winograd F(2x2,4x4) (csrc/wino44.hip; the discriminator's conv4, model/networks.py NLayerDiscriminator
 * `nn.Conv2d(ndf*4, ndf*8, kernel_size=4, stride=1, padding=1)`, and its data gradient): filter / input transform, ONE K-sliced launch
 * of the GEMM engine (25 positions), output transform.  in (NI,H,W,C) NHWC, out (NI, H+2*pad-3, W+2*pad-3, Cout), no bias / activation.
 * w is the layer's weight in PyTorch layout in BOTH directions: dgrad = 0: (Cout, C, 4, 4); dgrad = 1: `in` is dy, w is (C, Cout, 4, 4),
 * pad = 3 - the layer's padding, out = dx.  C % 16 == 0, Cout % 4 == 0, 16-byte aligned tensors, else RE2E_EUNSUPPORTED. */
size_t re2e_conv4x4_wino_workspace_bytes(int NI, int H, int W, int C, int Cout, int pad);
int re2e_conv4x4_wino(const float* in, int NI, int H, int W, int C, const float* w, int Cout, int pad, int dgrad, float* out,
                      void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* Its weight gradient the same way: gw (Cout, C, 4, 4) (+)= G^T [ sum_tiles (B^T d B) (x) (A dy A^T) ] G, the sum over tiles as a K-sliced
 * A^T B launch of the engine.  in (NI,H,W,C), dout (NI, H+2*pad-3, W+2*pad-3, Cout); beta 0 / 1; C % 4 == 0, Cout % 4 == 0. */
size_t re2e_conv4x4_wino_wgrad_workspace_bytes(int NI, int H, int W, int C, int Cout, int pad);
int re2e_conv4x4_wino_wgrad(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, int pad, float* gw, float beta,
                            void* workspace, size_t workspace_bytes, re2e_stream_t stream);
size_t re2e_conv_wgrad_workspace_bytes(int NI, int PH, int PW, int C, int Cout, int KH, int KW);
/* dW[Cout][C][KH][KW] = beta*dW + sum_pix dout[pix][co] * in[n][py*SY+kh+OY0][px*SX+kw+OX0][ci] */
int re2e_conv_wgrad(const float* in, int NI, int H, int W, int C, const float* dout, int Cout, int KH, int KW, int PH,
                    int PW, int SY, int SX, int OY0, int OX0, float* dW, float beta, void* workspace,
                    size_t workspace_bytes, re2e_stream_t stream);
/* dst[r][a][b][c] <- W[Cout][Cin][KH][KW] at tap (kh0+a*kstep, kw0+b*kstep); transpose=0: r=co,c=ci; 1: r=ci,c=co */
/* Stride-2 data gradient in one launch (all four output parity classes): dx[N][H][Wd][Cin] from
 * dz[N][OH][OW][Cout] and the PyTorch-layout weight W[Cout][Cin][KH][KW] (KH, KW even).
 * wt_ws: caller scratch of Cin*KH*KW*Cout floats (the per-class gathered weights). */
int re2e_conv_dgrad_s2(const float* dz, int N, int OH, int OW, int Cout, const float* W, int Cin, int KH, int KW, int H,
                       int Wd, int pad, float* dx, float* wt_ws, re2e_stream_t stream);
int re2e_conv_weight_gather(const float* W, float* dst, int Cout, int Cin, int KH, int KW, int transpose, int TA,
                            int TB, int kh0, int kw0, int kstep, re2e_stream_t stream);

/* ---- elementwise / layout helpers ------------------------------------------------------- */
/* out[d1][d0][:] = in[d0][d1][:]  (batch-first <-> time-major) */
int re2e_transpose01(const float* in, float* out, int D0, int D1, int W, re2e_stream_t stream);
/* y = act(x): stand-alone tanh / relu / lrelu(0.2) / sigmoid (the U-Net blocks' nn.LeakyReLU / nn.ReLU / nn.Sigmoid
 * modules, enhance_model.py:277-291; everywhere else the activation is a GEMM / conv epilogue) */
int re2e_act_fwd(const float* x, float* y, long n, int act, re2e_stream_t stream);
/* dz = dy * act'(y), y = activation OUTPUT (tanh / relu / lrelu / sigmoid); dz may alias dy */
int re2e_act_bwd(const float* dy, const float* y, float* dz, long n, int act, re2e_stream_t stream);
/* out[n] = beta*out[n] + sum_m A[m*lda+n]; workspace >= re2e_colsum_workspace_bytes */
size_t re2e_colsum_workspace_bytes(int M, int N);
int re2e_colsum(const float* A, int M, int N, long lda, float* out, float beta, void* workspace,
                size_t workspace_bytes, re2e_stream_t stream);
/* Activation backward fused with the bias gradient: dz = dy * act'(y) (rows x N, contiguous) and
 * out[N] = beta*out + column sums of dz, in one pass over dy/y.  Workspace as re2e_colsum. */
int re2e_act_bwd_colsum(const float* dy, const float* y, float* dz, int M, int N, int act, float* out, float beta,
                        void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* enhancer mask epilogue backward: dlin = dout * mix * m * (1-m)   (enhance_model.py:156-164) */
int re2e_mask_mul_bwd(const float* dout, const float* mix, const float* mask, float* dlin, long n,
                      re2e_stream_t stream);
/* the same with dlin's rows padded to ldd >= N floats (zeros): (rows, N) operands, (rows, ldd) result */
int re2e_mask_mul_bwd_ld(const float* dout, const float* mix, const float* mask, float* dlin, long rows, int N, int ldd,
                         re2e_stream_t stream);
/* out = a*b elementwise (clean*cos target of the mask-L1 loss, enhance_model.py:170) */
int re2e_mul(const float* a, const float* b, float* out, long n, re2e_stream_t stream);
/* CMVN: out[r][j] = (x[r][j] + c0[j]) * c1[j]; c0==NULL gives x*c1 (its backward)  (feat_model.py:132-134) */
int re2e_affine_cols(const float* x, const float* c0, const float* c1, float* out, long rows, int N,
                     re2e_stream_t stream);
/* Dropout with a counter-based mask (F.dropout model/e2e_ctc.py:51 -- always on, its default training=True --,
 * nn.LSTM(dropout=) between the layers of BLSTM model/e2e_encoder.py:156-157, nn.Dropout of the U-Net blocks
 * model/enhance_model.py:298): y_i = x_i / (1-p) when word (i&3) of Philox4x32-10(counter {i>>2, i>>34, call, 0}, key = seed)
 * is >= floor(p * 2^32), else 0.  The backward pass is the same call on dy with the same (seed, call): no mask is stored. */
int re2e_dropout(const float* x, float* y, long n, float p, unsigned long long seed, unsigned call, re2e_stream_t stream);
/* y = a*x + b*y elementwise (gradient accumulation) */
int re2e_axpby(float a, const float* x, float b, float* y, long n, re2e_stream_t stream);
/* dst[i][:] = src[idx[i]][:] (rows of width W); scatter is the inverse (dst[idx[i]][:] = src[i][:]) */
int re2e_gather_rows(const float* src, const int* idx_dev, float* dst, int nrows, int W, re2e_stream_t stream);
int re2e_scatter_rows(const float* src, const int* idx_dev, float* dst, int nrows, int W, re2e_stream_t stream);
/* zero rows t >= lens[b] of a (B,T,W) tensor: out = in masked (mask_by_length, e2e_common.py:190-195) */
int re2e_mask_rows(const float* in, float* out, const int* lens_dev, int B, int T, int W, re2e_stream_t stream);

/* ---- K1 batch pack/pad (data/mix_data_loader.py:264-302): ragged rows -> zero padded (B,Tmax,F) */
int re2e_pack_pad(const float* src_flat, const int* offsets_dev, const int* lens_dev, int B, int Tmax, int F,
                  float* dst, re2e_stream_t stream);

/* K1, input side (data/kaldi_io.py:376-456 + data/mix_data_loader.py:198-237): decode a batch of raw Kaldi matrix records
 * and pad them.  blob = the records' payload bytes (device); record b starts at blob + rec_off[b] and is
 *   kind 0 ('FM'): rows*F row-major fp32 (16-byte aligned start), or
 *   kind 2 ('CM'): kaldi CompressedMatrix format 1 -- {min f32, range f32, rows i32, cols i32}, F x 4 uint16
 *                  percentiles, F x rows uint8 column-major.
 * dst (B,Tmax,F) = decoded values, zero padded.  dst_log (optional) = (10*log10(max(x,1e-7)) + cmvn[0]) * cmvn[1]
 * (cmvn (2,F), optional); when dst_log is given dst is clamped at 1e-7 too, as the reference's in-place clamp does. */
int re2e_kaldi_decode_pad(const unsigned char* blob, const long* rec_off_dev, const int* kind_dev, const int* lens_dev, int B,
                          int Tmax, int F, float* dst, float* dst_log, const float* cmvn, re2e_stream_t stream);

/* ---- K2 fused fbank (model/feat_model.py:118-135): y = log(max((x^2) W, 1e-7)); optional
 * second output y_norm = (y + cmvn[0]) * cmvn[1]; optional third output pw_out = (x^2) W itself, which re2e_fbank_bwd takes
 * back.  W is given banded: for filter j the taps band_w[j*maxw + i] apply to bins band_off[j]+i, i<band_len[j]. */
int re2e_fbank_fwd(const float* x, long rows, int F, int NF, const int* band_off, const int* band_len,
                   const float* band_w, int maxw, float* y_raw, float* y_norm, const float* cmvn, float* pw_out,
                   re2e_stream_t stream);
/* dx = 2x * sum_j W[f][j] * (dy_raw + dy_norm*cmvn1)[j] / pw[j], zero where pw<=1e-7 (clamp).  pw (rows,NF) = the forward's pw_out.
 * Here the matrix is banded the other way round (the filters covering a bin are a contiguous run as well): bin f is covered by
 * filters bin_off[f] + i, i < bin_len[f], with weights bin_w[f*maxc + i]. */
int re2e_fbank_bwd(const float* x, long rows, int F, int NF, const int* bin_off, const int* bin_len, const float* bin_w, int maxc,
                   const float* pw, const float* dy_raw, const float* dy_norm, const float* cmvn, float* dx, re2e_stream_t stream);
/* Dense trainable filterbank (fbank_opti_type 'train', feat_model.py:105-109): x^2 W and its gradients are re2e_gemm calls;
 * these are the element-wise tail of feat_model.py:127-134: y = log(max(z,1e-7)) [-> (y + cmvn[0]) * cmvn[1]] and
 * dz = dy * cmvn[1] / z, zero where the clamp fired (cmvn (2,N) optional). */
int re2e_logclamp_fwd(const float* z, const float* cmvn, long rows, int N, float* y, re2e_stream_t stream);
int re2e_logclamp_bwd(const float* z, const float* cmvn, long rows, int N, const float* dy, float* dz, re2e_stream_t stream);
/* per-column sum and sum of squares over valid frames of y (B,T,NF): feat_model.py:62-80 on device */
int re2e_cmvn_stats(const float* y, const int* lens_dev, int B, int T, int NF, float* sum_out, float* sumsq_out,
                    re2e_stream_t stream);

/* ---- K10 reductions ---------------------------------------------------------------------- */
/* out[0] = mean_i loss(a_i - b_i) (b==NULL => constant `target`): F.mse_loss / l1 / smooth_l1
 * (joint_train.py:162-167) and GANLoss MSE-to-constant (gan_model.py:162-171) */
size_t re2e_reduce_workspace_bytes(long n);
int re2e_loss_fwd(const float* a, const float* b, float target, long n, int kind, float* out, void* workspace,
                  size_t workspace_bytes, re2e_stream_t stream);
/* da = beta*da + (*gscale_dev) * scale * dloss/da_i / n */
int re2e_loss_bwd(const float* a, const float* b, float target, long n, int kind, const float* gscale_dev,
                  float scale, float* da, float beta, re2e_stream_t stream);
/* out[0] = sum x_i^2 */
int re2e_sumsq(const float* x, long n, float* out, void* workspace, size_t workspace_bytes, re2e_stream_t stream);

/* ---- K5 pooling + VGG output packing (e2e_encoder.py:259-278) ------------------------------ */
/* 2x2/2 ceil-mode max pool over NHWC; idx_u8 stores the argmax (0..3) for the backward.  relu_in != 0: `in` is the output of the
 * ReLU of the convolution in front (e2e_encoder.py:260-261,264-265) and the pool's backward is to return the gradient of that
 * ReLU's INPUT: windows whose maximum is <= 0 get index 4, so re2e_maxpool2_bwd leaves their gradient out -- d(relu) costs no pass */
int re2e_maxpool2_fwd(const float* in, int NI, int H, int W, int C, float* out, unsigned char* idx_u8, int relu_in,
                      re2e_stream_t stream);
int re2e_maxpool2_bwd(const float* dout, const unsigned char* idx_u8, int NI, int H, int W, int C, float* din,
                      re2e_stream_t stream);
/* NHWC (NI,T,Fq,C) -> utterances [n_off, n_off+NI) of a time-major (T,NI_total,C*Fq) tensor with rows
 * t>=lens[n] zeroed (cut + re-pad); the offset lets two separately computed branches share one tensor */
int re2e_vgg_pack_fwd(const float* in, const int* lens_dev, int NI, int T, int Fq, int C, float* out, int NI_total,
                      int n_off, re2e_stream_t stream);
int re2e_vgg_pack_bwd(const float* dout, const int* lens_dev, int NI, int T, int Fq, int C, float* din, int NI_total,
                      int n_off, re2e_stream_t stream);

/* ---- K9 BatchNorm2d (train mode) + LeakyReLU(slope) over NHWC rows [P][C]: slope 0.2 = the discriminator's
 * BatchNorm2d + LeakyReLU(0.2) pairs (gan_model.py:76-88), slope 1.0 = plain BatchNorm2d (U-Net blocks,
 * enhance_model.py:281-296) */
size_t re2e_bn_workspace_bytes(long P, int C);
int re2e_bn_lrelu_fwd(const float* x, long P, int C, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float momentum, float eps, int train, float slope, float* y, float* save_mean,
                      float* save_invstd, void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* dy is the gradient w.r.t. the LeakyReLU output; dgamma/dbeta may be NULL (frozen D) */
int re2e_bn_lrelu_bwd(const float* dy, const float* x, long P, int C, const float* gamma, const float* beta,
                      const float* save_mean, const float* save_invstd, float slope, float* dx, float* dgamma, float* dbeta,
                      float gbeta, void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* Synchronised BatchNorm for data-parallel runs that shard ONE global batch (the same kernels, cut where the caller all-reduces):
 * re2e_bn_sync_partial: out[C] = sum over the LOCAL rows of (x - mean) (mean may be NULL) or of its square; the caller all-reduces and
 * divides by the global row count; re2e_bn_sync_finalize: global mean / biased variance -> save_mean, save_invstd, running statistics;
 * re2e_bn_apply: y = lrelu((x - mean) invstd gamma + beta); re2e_bn_sync_bwd_partial: out[2C] = local sum dz | sum dz*xhat;
 * re2e_bn_sync_bwd_apply: dx from the all-reduced sums scaled by P / Ptotal (the kernel divides by its own row count). */
int re2e_bn_sync_partial(const float* x, long P, int C, const float* mean, int square, float* out, void* workspace, size_t workspace_bytes,
                         re2e_stream_t stream);
int re2e_bn_sync_finalize(const float* mean, const float* var, long Ptotal, int C, float momentum, float eps, float* running_mean,
                          float* running_var, float* save_mean, float* save_invstd, re2e_stream_t stream);
int re2e_bn_apply(const float* x, long P, int C, const float* save_mean, const float* save_invstd, const float* gamma, const float* beta,
                  float slope, float* y, re2e_stream_t stream);
int re2e_bn_sync_bwd_partial(const float* dy, const float* x, long P, int C, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, float slope, float* out, void* workspace, size_t workspace_bytes, re2e_stream_t stream);
int re2e_bn_sync_bwd_apply(const float* dy, const float* x, long P, int C, const float* gamma, const float* beta, const float* save_mean,
                           const float* save_invstd, float slope, const float* sums, float* dx, re2e_stream_t stream);

/* ---- K4 bidirectional LSTM recurrence, packed-sequence semantics (nn.LSTM call sites
 * e2e_encoder.py:128-132,168-170).  Time-major.  xg_f/xg_r [T*B,4H] hold x W_ih^T + b_ih + b_hh
 * (gate order i,f,g,o) on entry and the ACTIVATED gates on exit (saved for the backward);
 * ybuf/cbuf are [(T+2)*B, 2H] with block 0 and block T+1 zero: y[t] lives in block t+1. */
size_t re2e_lstm_workspace_bytes(int B, int H);
/* Number of sequences (since the library was loaded) that a persistent recurrence kernel gave up on because a peer
 * workgroup never published its step (bounded spin; the outputs of such a sequence are NaN).  0 in a healthy run.
 * Synchronises the device; -1 if the counter cannot be read. */
int re2e_lstm_abort_count(void);
/* Test hooks (tests/test_dp_gpu.py, tests/test_trainers_gpu.py; no reference counterpart -- the reference is single-device and has no
 * persistent kernels).  re2e_debug_force_abort(n): the next n persistent FORWARD sequences behave as given up (outputs NaN, the
 * counter above rises) without running -- what JointTrainer.fit's recovery path is tested with.  re2e_debug_occupy: a kernel of
 * `workgroups` x 256 threads that holds `lds_bytes` of LDS each and spins for `usec` microseconds on `stream`: what a ring
 * all-reduce waiting for a slow peer looks like to the workgroup scheduler. */
/* Both hooks answer RE2E_EUNSUPPORTED unless RE2E_DEBUG_HOOKS=1 is in the environment (the tests set it). */
int re2e_debug_force_abort(int n);
int re2e_debug_occupy(int workgroups, int lds_bytes, int usec, re2e_stream_t stream);
/* The trainer's step gate (joint_train.py:188-193 plus the give-up protocol above), one 1-thread kernel and no host round trip.
 * base_dev: device int, the give-up count the trainer has acknowledged (ack != 0: set it to the current count first).  delta = count - *base_dev.
 * hold_next (optional): 1.0 when delta == 0, NaN otherwise -- the next step multiplies its losses by it, so a step enqueued before the host
 * repeated an aborted one applies nothing, on any replica.  stats_main (optional; the [norm, coef, finite | norm, 1, finite] of
 * re2e_clip_coef): finite := 0 and norm := NaN when delta != 0 or *extra_sumsq (optional: another optimizer's gradient sum of squares) is not
 * finite.  stats_d (optional): finite := 0 when delta != 0 or *d_requires (optional: another gate's finite flag) is 0.  delta_out (optional):
 * delta as a float, for the step's meters. */
int re2e_step_gate(int* base_dev, int ack, const float* extra_sumsq, float* stats_main, float* stats_d, const float* d_requires, float* hold_next,
                   float* delta_out, re2e_stream_t stream);
int re2e_lstm_seq_fwd(float* xg_f, float* xg_r, const float* whh_f, const float* whh_r, float* ybuf, float* cbuf,
                      const int* lens_dev, int T, int B, int H, void* workspace, size_t workspace_bytes,
                      re2e_stream_t stream);
/* g_f/g_r: activated gates in, d(pre-activation gates) out (in place).  dy [T*B,2H].
 * whh_*: W_hh [4H,H] as stored by nn.LSTM.  dc_state [B,2H] scratch (zeroed by the call).
 * dbias (optional, [2][4H]): the column sums of d(gates) over all T*B rows, forward direction then reverse = the gradient of
 * bias_ih and of bias_hh (autograd of nn.LSTM, e2e_encoder.py:128-132) -- accumulated beside the recurrence where the kernel
 * form allows it, so that the caller needs no pass of its own over the (T*B, 4H) tensors.
 * workspace (re2e_lstm_workspace_bytes) holds the MFMA-fragment-ordered weight / state copies. */
int re2e_lstm_seq_bwd(float* g_f, float* g_r, const float* whh_f, const float* whh_r, const float* dy,
                      const float* ybuf, const float* cbuf, float* dc_state, const int* lens_dev, int T, int B, int H,
                      float* dbias, void* workspace, size_t workspace_bytes, re2e_stream_t stream);

/* ---- K8 LSTMCell pointwise (decoder, e2e_decoder.py:131), embedding, cross-entropy -------- */
/* gates [B,4H] pre-activation in -> activated out; c_prev [B,H] -> c_out, h_out */
int re2e_lstm_cell_fwd(float* gates, const float* c_prev, float* c_out, float* h_out, int B, int H,
                       re2e_stream_t stream);
/* gates: activated in -> dgates (pre-activation) out; dh (+ dh2, optional: the decoder adds the direct gradient dZ[i] to
 * the carried one), dc_in -> dc_prev_out */
int re2e_lstm_cell_bwd(float* gates, const float* c_prev, const float* c_cur, const float* dh, const float* dh2,
                       const float* dc_in, float* dc_prev_out, int B, int H, re2e_stream_t stream);
/* One decoder step's LSTMCell (e2e_decoder.py:131) in one launch: gates [B,4D] (embedding half of the input projection + both
 * biases on entry) += ctx [B,E] W_ih[:, Dd:]^T + z_prev [B,D] W_hh^T (w_ctx = &W_ih[0][Dd], row pitch ldw), then the cell:
 * activated gates out, c_prev -> c_out, h_out.  E, D, ldw multiples of 4. */
int re2e_dec_gates_cell_fwd(const float* cx, const float* z_prev, const float* w_ctx, long ldw, const float* w_hh, float* gates,
                            const float* c_prev, float* c_out, float* h_out, int B, int E, int D, re2e_stream_t stream);
/* The whole teacher-forced decoder loop (e2e_decoder.py:113-152: AttLoc.forward e2e_attention.py:259-299 + LSTMCell per output token) as ONE
 * persistent launch (csrc/decloop.hip).  re2e_dec_loop_workspace_bytes returns 0 when the shape is outside the resident form's limits
 * (forward: B <= 32, D <= 320, E <= 512, D, E, A multiples of 4, C <= 12, T <= 2048; backward: E <= 512 and a multiple of 16, C <= 16,
 * 2*Fh+1 <= 255; both: every workgroup resident, one per CU -- csrc/decloop.hip dec_plan / dec_bwd_plan are the authority) or RE2E_DEC_PERSIST=0: the caller then runs the launch-per-step
 * sequence re2e_attloc_fwd + re2e_dec_gates_cell_fwd.  pre (B,T,A) = mlp_enc(enc), enc (B,T,E) masked encoder states, w_decT (D,A) = mlp_dec^T,
 * w_att (A,C), w_conv (C,2Fh+1), w_ctx = &W_ih[0][Dd] with row pitch ldw, w_hh (4D,D).  gates (L1,B,4D): embedding half of the input projection
 * + both biases on entry, activated gates on exit; z, c (L1+1,B,D) with block 0 = the initial state; w (L1,B,T), cx (L1,B,E),
 * conv (L1,B,T,C), dpj (L1,B,A): what re2e_attloc_bwd reads.  A give-up (bounded spin) is counted by re2e_lstm_abort_count. */
size_t re2e_dec_loop_workspace_bytes(int L1, int B, int T, int E, int D, int A, int C, int Fh);
int re2e_dec_loop_fwd(const float* pre, const float* enc, const int* hlens_dev, const float* w_decT, const float* w_att, const float* w_conv,
                      const float* gvec, const float* gvec_b, const float* w_ctx, long ldw, const float* w_hh, float* gates, float* z, float* c,
                      float* w, float* cx, float* conv, float* dpj, int L1, int B, int T, int E, int D, int A, int C, int Fh,
                      void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* The loop's backward as ONE persistent launch: the reverse of the above, token L1-1 .. 0.  On entry gates holds the activated gates saved by
 * the forward; on exit d(gates) (what the weight-gradient products read).  d_cx_all (L1,B,E), de_all (L1,B,T), ddp (L1,B,A) = d dec_proj and
 * d_pre (B,T,A; may be NULL) are written; the d conv rows of every token stay in the workspace for re2e_dec_loop_dwconv, which adds d W_conv to
 * partials (B, partial_floats) at float offset wconv_offset of each row (layout of re2e_attloc_partial_floats) -- a separate call so that it can
 * run on a weight-gradient stream.  dZ (L1,B,D): gradient of the decoder states.  0 bytes = shape outside the resident form (also
 * RE2E_DEC_PERSIST=0, or =2: forward only): the caller runs re2e_lstm_cell_bwd + re2e_gemm_skinny2 + re2e_attloc_bwd per token. */
size_t re2e_dec_loop_bwd_workspace_bytes(int L1, int B, int T, int E, int D, int A, int C, int Fh);
int re2e_dec_loop_bwd(const float* pre, const float* enc, const float* cx, const float* z, const float* c, const float* w, const float* conv,
                      const float* dpj, const float* dZ, const int* hlens_dev, const float* w_ctx, long ldw, const float* w_hh, const float* mlp_dec,
                      const float* w_att, const float* w_conv, const float* gvec, float* gates, float* d_cx_all, float* de_all, float* ddp,
                      float* d_pre, int L1, int B, int T, int E, int D, int A, int C, int Fh, void* workspace, size_t workspace_bytes,
                      re2e_stream_t stream);
int re2e_dec_loop_dwconv(const float* w, const int* hlens_dev, const void* workspace, size_t workspace_bytes, float* partials, int partial_floats,
                         int wconv_offset, int L1, int B, int T, int E, int D, int A, int C, int Fh, re2e_stream_t stream);
/* Two skinny products that share A (M <= 32 rows) in one launch: C1 = A[M,K] B1[K,N1], C2 = A B2[K,N2] (B row-major (K,N));
 * the decoder's backward step: d ctx = dgates W_ih[:, Dd:], d z = dgates W_hh. */
int re2e_gemm_skinny2(int M, int K, const float* A, long lda, const float* B1, long ldb1, int N1, float* C1, long ldc1,
                      const float* B2, long ldb2, int N2, float* C2, long ldc2, re2e_stream_t stream);
int re2e_embedding_fwd(const float* table, const int* ids_dev, int n, int D, float* out, long ldo,
                       re2e_stream_t stream);
/* dtable[v][:] = beta*dtable + sum_{i: ids[i]==v} dout[i][:] in index order (deterministic) */
int re2e_embedding_bwd(const float* dout, long ldo, const int* ids_dev, int n, int D, int V, float* dtable,
                       float beta, re2e_stream_t stream);
/* Row-wise arg-max (lowest index on ties) of x[R][V] (leading dimension ldx): the token that scheduled
 * sampling and Decoder.calculate_all_attentions feed back (e2e_decoder.py:123-127, :408-412 `y_i.topk(1)`). */
int re2e_argmax_rows(const float* x, int R, int V, long ldx, int* out_ids, re2e_stream_t stream);
/* out[R][V] = log_softmax(x[R][V]) row-wise (F.log_softmax of Decoder.recognize_beam e2e_decoder.py:263 and
 * CTC.log_softmax e2e_ctc.py:68-75) */
int re2e_log_softmax_rows(const float* x, int R, int V, long ldx, float* out, re2e_stream_t stream);

/* F.cross_entropy(ignore_index=-1, mean) * scale and th_accuracy (e2e_decoder.py:155-161).
 * out[0]=loss, out[1]=#correct, out[2]=#valid ; lse [R] saved for the backward */
int re2e_ce_fwd(const float* logits, const int* targets_dev, int R, int V, float scale, float* out, float* lse,
                void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* dlogits = (*gscale_dev)*scale/#valid * (softmax - onehot) on valid rows, 0 on ignored rows */
int re2e_ce_bwd(const float* logits, const int* targets_dev, const float* lse, const float* fwd_out, int R, int V,
                float scale, const float* gscale_dev, float* dlogits, re2e_stream_t stream);
/* Label-smoothing regulariser of the decoder loss (e2e_decoder.py:162-166):
 * out[0] = -(1/nutt) * sum over ALL R rows and V labels of log_softmax(logits)[r][v] * dist[v]; workspace R floats.
 * bwd: dlogits[r][v] = gscale[0] * (softmax[r][v] * sum(dist) - dist[v]) / nutt. */
int re2e_lsm_fwd(const float* logits, const float* dist, int R, int V, int nutt, float* out, void* workspace,
                 size_t workspace_bytes, re2e_stream_t stream);
int re2e_lsm_bwd(const float* logits, const float* dist, int R, int V, int nutt, const float* gscale, float* dlogits,
                 re2e_stream_t stream);

/* ---- K6 CTC (warp-ctc call site e2e_ctc.py:63): logits (T,B,V) raw activations, blank 0,
 * loss = sum_b nll_b / B.  labels_dev: flat int32; label_off_dev/label_len_dev per utterance. */
size_t re2e_ctc_workspace_bytes(int T, int B, int Lmax);
int re2e_ctc_fwd(const float* logits, int T, int B, int V, const int* hlens_dev, const int* labels_dev,
                 const int* label_off_dev, const int* label_len_dev, int Lmax, float* loss_out, float* nll_per_utt,
                 void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* dlogits = (*gscale_dev)/B * (softmax - occupancy) for t<hlens[b], 0 otherwise; uses the workspace of fwd.  dlogits has rows of
 * ldd >= V floats (columns V .. ldd-1 are written as zeros): an odd vocabulary (V = 4233) padded to a multiple of 16 lets the two
 * products of the projection's backward run on the engine's 16-byte-load path. */
int re2e_ctc_bwd(const float* logits, int T, int B, int V, const int* hlens_dev, const int* labels_dev,
                 const int* label_off_dev, const int* label_len_dev, int Lmax, const float* nll_per_utt,
                 const float* gscale_dev, float* dlogits, int ldd, const void* workspace, re2e_stream_t stream);

/* ---- N3 CTC prefix scores for joint CTC/attention beam search (CTCPrefixScore model/e2e_ctc.py:78-155 as called per
 * hypothesis at model/e2e_decoder.py:231-263), all `nh` live hypotheses of one output position in one launch.
 * lpz (T,V) CTC log posteriors; att_lsm (nh,V) attention log-probabilities; r_prev (nh,T,2) forward variables of each
 * hypothesis; last_label / out_len (labels without <sos>) / prev_score (nh).  Per hypothesis: cand_out (ctc_beam) = its top
 * labels in torch.topk order, ctc_score_out = log prefix probabilities, local_out = att_weight*att + ctc_weight*(ctc - prev), r_new
 * (nh,ctc_beam,T,2) = the candidates' new forward variables.  ctc_beam <= 64 (RE2E_EUNSUPPORTED beyond). */
int re2e_ctc_prefix_score(const float* lpz, int T, int V, const float* att_lsm, int nh, const float* r_prev, const int* last_label_dev,
                          const int* out_len_dev, const float* prev_score_dev, int ctc_beam, float att_weight, float ctc_weight, int blank,
                          int eos, int* cand_out, float* local_out, float* ctc_score_out, float* r_new, re2e_stream_t stream);
/* The same scores for a caller-given candidate list cand_dev (nh, ncand) -- ctc_weight == 1.0 scores all V labels in the order of their
 * attention scores (a device-side stable sort): one thread per candidate, outputs (nh, ncand) in the list's order, r_new (nh*ncand, T, 2). */
int re2e_ctc_prefix_score_cands(const float* lpz, int T, int V, const float* att_lsm, int nh, const float* r_prev, const int* last_label_dev,
                                const int* out_len_dev, const float* prev_score_dev, const int* cand_dev, int ncand, float att_weight,
                                float ctc_weight, int blank, int eos, float* local_out, float* ctc_score_out, float* r_new, re2e_stream_t stream);

/* ---- K7 location-aware attention step (model/e2e_attention.py:258-297) -------------------- */
/* per utterance b: w = softmax_t(2*(gvec . tanh(W_att conv(att_prev) + pre[b,t] + W_dec z[b]) + gb)),
 * c[b] = sum_t w[t]*enc[b,t].  att_prev==NULL => uniform 1/hlen over valid frames.
 * w_decT is mlp_dec.weight TRANSPOSED, (dunits, adim).  conv_out (B,T,chans) and dp_out (B,adim) are the
 * location-conv output and W_dec z, saved for the backward; e_scratch (B,T) is scratch.
 * The work is cut into 32-frame workgroups (grid = ceil(T/32) x B) + a per-utterance softmax/context pass. */
int re2e_attloc_fwd(const float* pre, const float* enc, const float* z, const float* att_prev, const int* hlens_dev,
                    const float* w_decT, const float* w_att, const float* w_conv, const float* gvec, const float* gvec_b,
                    int B, int T, int eprojs, int dunits, int adim, int chans, int filts, float* w_out, float* c_out,
                    long ldc_out, float* conv_out, float* dp_out, float* e_scratch, re2e_stream_t stream);
size_t re2e_attloc_partial_floats(int adim, int chans, int filts);
size_t re2e_attloc_workspace_bytes(int B, int T, int adim, int chans);
/* backward of one step: writes de_out (B,T) = d(energy) of this step (kept by the caller for re2e_attloc_dpre),
 * d_att_prev (may be NULL for step 0) and d_decproj [B,adim] (for the mlp_dec GEMMs), accumulates the location-filter
 * part of the per-utterance weight-grad partials (+=), which are laid out
 * [B][gvec(adim) | gvec_b(1) | w_att(adim*chans) | w_conv(chans*(2*filts+1))].  cx_in = the forward context
 * c (B,eprojs), conv_in / dp_in = the tensors saved by re2e_attloc_fwd.  Neither the encoder-state gradient nor
 * d_pre / dgvec / dW_att are produced here: call re2e_attloc_denc and re2e_attloc_dpre once after the loop. */
int re2e_attloc_bwd(const float* pre, const float* enc, const float* att_prev, const float* w_cur, const int* hlens_dev,
                    const float* w_att, const float* w_conv, const float* gvec, const float* conv_in, const float* dp_in,
                    const float* cx_in, const float* dc, long ld_dc, const float* dw_in, int B, int T, int eprojs, int adim,
                    int chans, int filts, float* de_out, float* d_att_prev, float* d_decproj, float* partials,
                    void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* after the loop: d_pre (B,T,adim) = sum over the L1 steps of d(pre-activation) (overwritten), and the gvec / gvec_b /
 * w_att parts of the partials (+=), recomputed from pre, conv_all (L1,B,T,chans), dp_all (L1,B,adim) (the tensors saved
 * by re2e_attloc_fwd, stacked over the steps) and de_all (L1,B,T) (written by re2e_attloc_bwd). */
int re2e_attloc_dpre(const float* pre, const float* conv_all, const float* dp_all, const float* de_all, const float* w_att,
                     const float* gvec, int L1, int B, int T, int adim, int chans, int filts, float* d_pre, float* partials,
                     void* workspace, size_t workspace_bytes, re2e_stream_t stream);
/* d_enc[b,t,:] = beta*d_enc + sum_i w_all[i,b,t] * dc_all[i,b,:]  (w_all (L1,B,T), dc_all (L1,B,eprojs)) */
int re2e_attloc_denc(const float* w_all, const float* dc_all, int L1, int B, int T, int eprojs, float* d_enc, float beta,
                     re2e_stream_t stream);

/* ---- K11 optimizer (joint_train.py:131-140,188-193; Appendix A.16) ------------------------ */
/* stats[0]=||g||_2, stats[1]=clip coefficient (<=1), stats[2]=1 if finite else 0; from sumsq[0];
 * stats[3..5] = the same with coefficient 1 (NaN gate only).  stats_dev holds 6 floats. */
int re2e_clip_coef(const float* sumsq_dev, float max_norm, float* stats_dev, re2e_stream_t stream);
/* g <- coef*g applied on the fly; skipped entirely when stats[2]==0 (NaN guard, no host sync) */
int re2e_adadelta_step(float* p, const float* g, float* sq_avg, float* acc_delta, long n, float rho, float eps,
                       float lr, const float* stats_dev, re2e_stream_t stream);
int re2e_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                   float eps, int step, const float* stats_dev, re2e_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RE2E_H */
