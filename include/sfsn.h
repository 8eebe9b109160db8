/*
 * sfsn.h -- C ABI of libsfsn_hip.so: the MI355X (gfx950) implementation of Spiking-FullSubNet's
 * recurrent inference hot path (full-band + stacked sub-band Gated-Spiking-Neuron scans with the
 * feature prologue and deep-filter epilogue around them).
 *
 * The reference (haoxiangsnr/spiking-fullsubnet) is pure Python and has no FFI of its own; the seam this
 * ABI fills is the one SURVEY.md 8b defines: everything an `nn.Module.forward` of the reference does
 * between `stft()` and `istft()`.  Each entry point names the reference lines it replaces (paths relative
 * to the reference root):
 *     NEURON = audiozen/models/spiking_fullsubnet/efficient_spiking_neuron.py
 *     MODEL  = audiozen/models/spiking_fullsubnet/modeling_spiking_fullsubnet.py
 *     FROZEN = recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq.py
 *
 * Conventions
 *   - plain pointers, ints and floats only; no torch / HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*, NULL = the null stream);
 *   - every pointer named in a launch is DEVICE memory owned by the caller; the library never allocates,
 *     frees or retains caller memory and keeps no global state: all entry points are re-entrant;
 *   - launches are asynchronous on `stream`; the return value reports argument / launch errors only;
 *   - layouts are the reference's: time-major [T][R][feat] inside a sequence model, [B][F][T] for
 *     spectra, interleaved (re, im) floats for complex64;
 *   - return codes: SFSN_OK or a negative SFSN_E* value; sfsn_strerror() names them.
 *
 * There is no CPU implementation behind this ABI and no fallback: without a gfx950 device every launch
 * returns SFSN_EHIP.
 */
#ifndef SFSN_H
#define SFSN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFSN_ABI_VERSION 19 /* bumped on every struct / signature change: a stale .so must not load */

#define SFSN_OK 0
#define SFSN_EINVAL (-1)       /* malformed argument (NULL where required, size <= 0, misaligned pointer)      */
#define SFSN_EUNSUPPORTED (-2) /* valid request the kernels do not cover (e.g. hidden size > 320)               */
#define SFSN_EHIP (-3)         /* HIP runtime error at launch (no device, invalid pointer, ...)                 */
#define SFSN_EDIVISIBLE (-4)   /* band width not divisible by the centre size: the reference's ValueError       */

#define SFSN_MAX_SEGMENTS 8    /* independent row segments (sub-band groups) per grouped launch                 */
#define SFSN_MAX_HIDDEN 320    /* largest hidden size held register-resident (baseline_m/l/xl full-band)        */
#define SFSN_MAX_GROUPS 8      /* sub-band groups per model                                                     */

int sfsn_abi_version(void);
/* First 16 hex digits of the sha256 of the sources this library was built from (include/sfsn.h and the files under csrc/): lets a binding
 * refuse a stale build whose struct layouts or entry points no longer match the header it was written against. */
const char* sfsn_source_hash(void);
const char* sfsn_strerror(int code);
/* Number of visible HIP devices (0 when there is none); never fails. */
int sfsn_device_count(void);

/* ------------------------------------------------------------------------------------------------------
 * Weight packing (HOST function, pure integer/bit work, no device needed).
 *
 * The recurrent product h.W_hh^T (NEURON:143), the layer>=2 input product S.W_ih^T (NEURON:141 with a
 * spike input) and the projection S.W_p^T (MODEL:118, FROZEN:125) all have a BINARY left operand
 * (spikes are exactly 0.0/1.0, NEURON:89).  They run on the int8 matrix cores: each fp32 weight row n is
 * split into three signed base-256 digits of a 24-bit fixed-point value with a per-row power-of-two scale,
 *     W[n][k] ~= (d2*65536 + d1*256 + d0) * dq[n],   dq[n] = 2^e[n] * 2^-23,  |W - W~| <= dq[n]/2,
 * (i.e. at most one fp32 ulp of the row's largest binade), the three int32 accumulators are exact, and
 * their recombination rounds once -- more accurate than, and independent of the summation order of, any
 * fp32 GEMM.  The digits are stored in MFMA A-fragment order for v_mfma_i32_16x16x64_i8:
 *     packed[d][nt][ks][lane][16]  with n = nt*16 + (lane & 15),  k = ks*64 + (lane >> 4)*16 + byte,
 * zero padded to NT = ceil(n_out/16) row tiles and KS = ceil(k_in/64) k-steps.
 * ---------------------------------------------------------------------------------------------------- */
size_t sfsn_w3_packed_bytes(int n_out, int k_in);               /* 3 * NT * KS * 1024 */
int sfsn_w3_padded_rows(int n_out);                            /* NT * 16 (length of dq) */
int sfsn_w3_pack(const float* w /* [n_out][k_in] host */, int n_out, int k_in, int8_t* packed /* host */,
                 float* dq /* [NT*16] host */);
/* The 16-bit-weight fast mode (BASELINE configs[2] "bf16" wording; SURVEY 0: the parity gate is the exact mode, this one reports
 * its spike agreement and output error separately): bits = 16 rounds every weight to 16 significant bits of the same per-row
 * grid and leaves the least-significant digit plane all zero; bits = 24 is sfsn_w3_pack.  Same layout, same kernels. */
int sfsn_w3_pack_bits(const float* w /* host */, int n_out, int k_in, int bits /* 24 | 16 */, int8_t* packed /* host */,
                      float* dq /* host */);
/* Inverse (tests): reconstruct W~ [n_out][k_in] from the packed digits. */
int sfsn_w3_unpack(const int8_t* packed, const float* dq, int n_out, int k_in, float* w);

/* ------------------------------------------------------------------------------------------------------
 * GSN layer scan -- replaces GSULayer.forward (NEURON:75-81: the python loop over T), GSUCell.forward
 * (NEURON:132-153) and Triangle.forward (NEURON:84-92), for several independent row segments at once
 * (the sub-band groups share H, MODEL:239-261).
 *
 * Per segment and per frame t:
 *     pre_f = zin[t][r][j] + (h . W_hh^T)[j]                               NEURON:140-145
 *     pre_g = zin[t][r][H + j] + (h . W_hh^T)[H + j]                       (unshared weights)
 *           = pre_f + (bias[H + j] - bias[j])                              (shared weights: one product serves both gates)
 *     f = sigmoid(pre_f);  c' = f*c + (1-f)*pre_g;  c'' = fma(c', bn_alpha[j], bn_beta[j])
 *     h' = (c'' >= 0);  carry (h', c'')                                    NEURON:146-153
 * zin is the time-parallel INPUT TERM INCLUDING bias_ih, as the reference associates it ((x.W_ih^T + bias) + h.W_hh^T):
 *     shared:   zin[.][j] = (x . W_ih^T)[j] + bias[j]            (forget-gate bias; the scan adds the bias difference)
 *     unshared: zin[.][g*H + j] = (x . W_ih^T)[g*H + j] + bias[g*H + j]
 * precomputed by sfsn_input_proj_f32 / sfsn_spike_proj with their bias argument;  bn_alpha /
 * bn_beta are eval-mode BatchNorm1d folded the way ATen's CPU kernel evaluates it (alpha = gamma /
 * sqrt(var + eps), beta = fma(-mean, alpha, bias); identity = (1, 0) when bn=False).
 * The hidden state h lives in LDS as int8, the membrane c in registers, W_hh in registers as packed int8
 * digits for the whole scan; one workgroup owns 16 rows.  All segments of one launch must request the same set
 * of optional outputs (it selects the compiled kernel variant).  T is the number of frames of THIS launch: a long
 * sequence may be scanned in chunks (pointers advanced by the caller, h_state / c_state carrying the state), which is
 * how independent layers / models are pipelined over time on separate streams, and how streaming inference works.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct sfsn_scan_segment {
    const float* zin;      /* [T][R][G*H] input term incl. bias (see above), G = 1 shared / 2 unshared      */
    const int8_t* w_hh;    /* sfsn_w3_pack(W_hh [G*H][H])                                                     */
    const float* w_dq;     /* [pad16(G*H)] from sfsn_w3_pack                                                  */
    const float* bias;     /* [2H]  bias_ih                                                                   */
    const float* bn_alpha; /* [H]                                                                             */
    const float* bn_beta;  /* [H]                                                                             */
    float* h_state;        /* [R][H] in: h at t=-1 (0/1), out: h at t=T-1   (NEURON:50-62 state in/out)       */
    float* c_state;        /* [R][H] in/out membrane                                                          */
    float* spikes_f32;     /* [T][R][H] out, nullable  (the reference's all_layer_outputs entry)              */
    int8_t* spikes_i8;     /* [T][R][pad64(H)] out, REQUIRED (B operand of the next sfsn_spike_proj)         */
    float* membrane;       /* [T][R][H] out, nullable  (post-BN membrane; parity tests only; needs spikes_f32)*/
    int R;                 /* rows in this segment (> 0)                                                      */
    unsigned long long* spike_count; /* nullable, 8-byte aligned (ABI 16).  When spikes_f32 is NULL the launch ADDS the number
                            * of spikes it wrote for this segment (rows < R, neurons < H, its T frames) -- the reduction
                            * audiozen/metric.py:303-340 takes of an all_layer_outputs entry, formed inside the scan:
                            * zero it once per forward, chunked calls accumulate.  Ignored when spikes_f32 is given. */
} sfsn_scan_segment;

int sfsn_gsn_layer_scan(const sfsn_scan_segment* segs /* host array */, int n_segs, int T, int H, int shared,
                        int rows_per_wg /* 16, 8, 4, or 0 = choose so that the launch covers ~all CUs */, void* stream);

/* The same scan for weights packed with 16 bits (sfsn_w3_pack_bits(.., 16): digit plane 0 is zero): the zero plane's matrix
 * instructions are skipped (8 instead of 12 per tile and step; the sums are the same, so the results equal sfsn_gsn_layer_scan's
 * on the same packed weights).  Covers what the IO-specialised scan covers (shared gates, H <= 224, 4 or 8 rows per workgroup, no
 * membrane output); SFSN_EUNSUPPORTED otherwise -- the caller then uses sfsn_gsn_layer_scan.  BASELINE configs[2]'s 16-bit mode. */
int sfsn_gsn_layer_scan_w16(const sfsn_scan_segment* segs /* host */, int n_segs, int T, int H, int shared, int rows_per_wg, void* stream);

/* Separate gate weights that do not fit one compute unit (shared = 0, H > 256: baseline_xl's full-band model, 2 x 320 x 320 x 3 bytes):
 * sfsn_gsn_layer_scan streams all of W_hh from the L2 every step (10 us per step); here the neuron tiles of a 16-row block are split
 * over several workgroups that keep their share resident in LDS and exchange the new spikes every step through `scratch`
 * (sfsn_scan_split_scratch_bytes(R, H) bytes, 16-byte aligned, ZEROED by the caller before every call; its first word is a sticky
 * error word: non-zero = a bounded wait expired, the outputs are invalid).  Same results as sfsn_gsn_layer_scan, bit for bit.  One
 * segment; ceil(R / 16) x splits workgroups must be co-resident (<= compute units): SFSN_EUNSUPPORTED otherwise, and for shapes
 * sfsn_gsn_layer_scan serves from one compute unit.  (ABI 17) */
size_t sfsn_scan_split_scratch_bytes(int R, int H);
int sfsn_gsn_layer_scan_split(const sfsn_scan_segment* segs /* host */, int n_segs, int T, int H, int shared, void* scratch,
                              size_t scratch_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Training-mode cell steps -- replace one iteration of GSULayer.forward's loop (NEURON:78-80) around
 * GSUCell.forward (NEURON:132-153) with nn.BatchNorm1d in TRAINING mode (batch statistics of this step over all R rows, running
 * statistics updated with `momentum`, NEURON:123,149-150), and the matching step of the backward pass through Triangle.backward
 * (NEURON:94-101).  One launch per time step: the rows of a step are coupled by the normalisation.  All tensors fp32, device.
 *   z [R][G*H]: x_t . W_ih^T WITHOUT bias (G = 1 shared: one product serves both gates; 2: forget | cell);  bias [2H];
 *   w_hh [G*H][H] fp32;  h_prev [R][H] spikes (0 / 1: anything non-zero counts as 1), c_prev [R][H];  bn_w / bn_b [H] (both NULL: bn = False);  running_mean / running_var [H]
 *   nullable (not updated).  Outputs of the forward step (saved for the backward step): spikes = h_t, u = c_t (post-BN
 *   membrane), xhat (normalised, pre-affine; bn only), f (forget gate), g (pre-activation of the cell gate), invstd [H] (bn only).
 * Backward step: dh_up (gradient w.r.t. h_t from above, nullable), dh_rec (gradient w.r.t. h_t through step t+1's recurrent
 *   product, nullable), dc_next (gradient w.r.t. c_t from step t+1, nullable) -> d_gates [R][2H] (d pre_f | d pre_g), d_z [R][G*H]
 *   (shared: their sum = the gradient of the shared product; unshared: pass NULL and use d_gates), dc_prev [R][H], and
 *   d_bn_w / d_bn_b [H] ACCUMULATED (+=).  The recurrent part of dL/dh_t is either handed in (dh_rec) or formed by the step itself
 *   from the previous launch's d_z (dz_next . W_hh: no GEMM launch between the steps).
 * Rows per step beyond ~64 are spread over several workgroups per neuron tile, which exchange their partial sums through `scratch`
 * (sfsn_train_scratch_bytes(H) bytes, ZEROED by the caller before the first step of a layer call: 8-byte {value, epoch} granules)
 * and wait for each other inside the launch; `epoch` = 1, 2, ... counts the steps issued on that scratch buffer (forward and backward use their own buffers). */
size_t sfsn_train_scratch_bytes(int H);
/* SFSN_OK when BOTH step kernels take this geometry (their LDS needs differ): call once per layer before the first forward step. */
int sfsn_gsn_train_check(int R, int H, int shared);
/* The same question for the per-step launches alone (their own, smaller LDS needs; with several row blocks per neuron tile also that
 * a step launch can hold its workgroups resident: needs the device then).  training.py asks it before it falls back from the one-launch
 * layer call (sfsn_gsn_train_multi_check said SFSN_EUNSUPPORTED) to a launch per step, and for the eval-mode-BatchNorm path.  ABI 16. */
int sfsn_gsn_train_step_check(int R, int H, int shared);
int sfsn_gsn_train_step_fwd(const float* z, const float* w_hh, const float* bias, const float* h_prev, const float* c_prev,
                            const float* bn_w, const float* bn_b, float* running_mean, float* running_var, float momentum, float eps,
                            int R, int H, int shared, float* spikes, float* u, float* xhat, float* f, float* g, float* invstd,
                            void* scratch, unsigned epoch, void* stream);
int sfsn_gsn_train_step_bwd(const float* dz_next /* [R][G*H] d_z of step t+1, nullable */, const float* w_hh /* [G*H][H] */,
                            const float* dh_up, const float* dh_rec, const float* dc_next, const float* u, const float* xhat,
                            const float* f, const float* g, const float* c_prev, const float* invstd, const float* bn_w, int R, int H,
                            int shared, float* d_gates, float* d_z, float* dc_prev, float* d_bn_w, float* d_bn_b, void* scratch,
                            unsigned epoch, void* stream);

/* A whole layer call in ONE launch (ABI 14; ABI <= 13 enqueued T step launches): forward (t = 0 .. T-1) or backward (t = T-1 .. 0).
 * The workgroups stay resident for all T steps -- weight tile, carried membrane (forward) / its gradient (backward) in LDS -- and
 * exchange what a step needs from other workgroups through the L2 (csrc/sfsn_train.hip: packed spikes / d_z written through, one
 * publish counter per row block; the BatchNorm partial sums as for the step entries).  Arithmetic is the step entries', value for
 * value.  Tensors as for the step entries with a leading [T]: z [T][R][G*H]; spikes, u, xhat, f, g, dh_up [T][R][H]; invstd [T][H];
 * d_gates [T][R][2H]; d_z [T][R][H] (shared; NULL otherwise).  Zero initial state (MODEL:100-106).  `zero` and `dc_work` are unused
 * (kept for the ABI-13 call shape; may be NULL).  `scratch`: sfsn_train_seq_scratch_bytes(R, H) bytes, ZEROED by the caller before the
 * call, one buffer per call (forward and backward use their own); its last four 32-bit words hold the error word as before. */
size_t sfsn_train_seq_scratch_bytes(int R, int H);
/* Several layer calls of the same (T, H, gate sharing) in ONE launch per direction -- the sub-band groups of a model are independent
 * of one another, and a layer call leaves most of the chip idle: their workgroups sit side by side in one grid.  Per call the
 * tensors of sfsn_gsn_train_seq_fwd / _bwd and its own zeroed scratch buffer.  n <= 8.  All calls with BatchNorm or all without.
 * SFSN_EUNSUPPORTED when the launch could not hold every call's workgroups resident together (sfsn_gsn_train_multi_check says so
 * up front; it needs the device): issue the calls separately then.  A launch with more calls than fit at the single call's geometry
 * (~160 workgroups per call) gives ALL its calls fewer, larger row blocks, per direction, until it fits or a block's rows exceed the
 * LDS (round 5: the layers of several stacks side by side, training.GSNStackTrainFn); the BatchNorm partial sums are then merged in
 * another blocking (last-bit differences against the single call). */
typedef struct {
    const float *z, *w_hh, *bias, *bn_w, *bn_b;
    float *running_mean, *running_var;
    float momentum, eps;
    int R;
    float *spikes, *u, *xhat, *f, *g, *invstd;
    void* scratch;
    /* ABI 17: a layer call cut into CHUNKS of frames (so that layer l + 1 can work on chunk c while layer l works on chunk c + 1 in the
     * same launch: two calls of one multi launch).  h0 / c0 [R][H], both or neither: the spikes and the (post-BatchNorm) membrane of the
     * frame before this call's first -- rows `spikes[-1]` / `u[-1]` of the previous chunk's call; NULL = zero state.  The running
     * statistics continue through the calls in launch order; momentum < 0 counts from the batches tracked before THIS call. */
    const float *h0, *c0;
    int T; /* > 0: this call's own frame count (chunks of unequal length side by side); 0: the launch's T */
} SfsnTrainSeqFwd;
typedef struct {
    const float *w_hh, *dh_up, *u, *xhat, *f, *g, *invstd, *bn_w;
    int R;
    float *d_gates, *d_z, *d_bn_w, *d_bn_b;
    void* scratch;
    /* ABI 17, chunked calls (issued last chunk first): dc_in [R][H] = dL/dc carried out of the chunk BEHIND this one (its dc_out); with it
     * the tensors must continue behind this call's T frames -- d_z / d_gates frame T is that chunk's first, already written.  dc_out
     * [R][H], nullable: dL/dc carried out of this call's first frame.  has_prev: `u` has a frame before this call's first (u - R*H floats:
     * the membrane the first step's forget gate multiplied), i.e. this is not the sequence's first chunk. */
    const float* dc_in;
    float* dc_out;
    int has_prev;
    int T; /* > 0: this call's own frame count; 0: the launch's T */
} SfsnTrainSeqBwd;
int sfsn_gsn_train_multi_check(const int* R, int n, int H, int shared);
int sfsn_gsn_train_seq_fwd_multi(const SfsnTrainSeqFwd* calls, int n, int T, int H, int shared, void* stream);
int sfsn_gsn_train_seq_bwd_multi(const SfsnTrainSeqBwd* calls, int n, int T, int H, int shared, void* stream);
int sfsn_gsn_train_seq_fwd(const float* z, const float* w_hh, const float* bias, const float* bn_w, const float* bn_b,
                           float* running_mean, float* running_var, float momentum, float eps, int T, int R, int H, int shared,
                           const float* zero, float* spikes, float* u, float* xhat, float* f, float* g, float* invstd, void* scratch,
                           void* stream);
int sfsn_gsn_train_seq_bwd(const float* w_hh, const float* dh_up, const float* u, const float* xhat, const float* f, const float* g,
                           const float* invstd, const float* bn_w, int T, int R, int H, int shared, const float* zero, float* d_gates,
                           float* d_z, float* dc_work, float* d_bn_w, float* d_bn_b, void* scratch, void* stream);

/* Fused-input variant for layers >= 1 (their input is the previous layer's spikes): the input term is computed inside
 * the scan from the int8 spikes and the packed input weights, so that sfsn_spike_proj's [T][R][H] fp32 result never makes
 * its round trip through HBM.  Bit-identical to sfsn_spike_proj(bias = bias[0:H]) + sfsn_gsn_layer_scan.  Shared gate
 * weights, 128 < H <= 256, 16 rows per workgroup (the geometry for a full chip); `zin` and `membrane` of the segments are
 * ignored / must be NULL.  SFSN_EUNSUPPORTED for other shapes: the caller then uses the two-call form. */
typedef struct sfsn_fused_input {
    const int8_t* spikes_in; /* [T][R][pad64(H)] the previous layer's spikes_i8                           */
    const int8_t* w_ih;      /* sfsn_w3_pack(W_ih [H][H])                                                  */
    const float* w_ih_dq;    /* [pad16(H)]                                                                 */
} sfsn_fused_input;

int sfsn_gsn_layer_scan_fused(const sfsn_scan_segment* segs /* host */, const sfsn_fused_input* fin /* host, one per segment */,
                              int n_segs, int T, int H, void* stream);

/* Layer-0 twin: the real-valued input product inside the scan (bf16 3-way split, bit-identical to sfsn_input_proj_f32 +
 * sfsn_gsn_layer_scan).  Same restrictions as above plus: I even, I <= 64, R % 16 == 0, x 16-byte aligned.  All segments of
 * a launch see the same LDS layout only if they share I: the slot size is taken from each segment's own I. */
typedef struct sfsn_fused_x {
    const float* x;    /* [T][R][I] the layer input (sfsn_features' output) */
    const float* w_ih; /* [H][I] fp32 input weights, row-major              */
    int I;
} sfsn_fused_x;

int sfsn_gsn_layer_scan_fused_x(const sfsn_scan_segment* segs /* host */, const sfsn_fused_x* fin /* host, one per segment */,
                                int n_segs, int T, int H, void* stream);

/* Layer-pipelined stack scan -- replaces StackedGSU.forward (NEURON:50-62) for ALL layers of a stack in ONE launch.
 * The reference runs layer l over all T frames, then layer l+1 (NEURON:56-61); layer l+1 needs frame t of layer l only at
 * frame t, so here every layer's rows get their own workgroups (weights resident for the whole launch) and the workgroups of
 * layer l+1 trail their producers of layer l by a few frames: int8 spikes handed over through L2 with write-through stores and
 * per-workgroup progress counters.  The critical path of a stack is one chain of T steps instead of n_layers chains, and the
 * fp32 input term of layers >= 1 makes no round trip through HBM.  Bit-identical to sfsn_input_proj_f32 / sfsn_spike_proj +
 * sfsn_gsn_layer_scan per layer.
 *   segs[l * n_segs + i]: layer l of segment i, as for sfsn_gsn_layer_scan; `zin` is read for layer 0 (the input term incl.
 *       bias, written before the launch).  For H > 256 (the two matrices of a layer >= 1 do not fit one CU) `zin` of layers >= 1
 *       must point at a [T][R][H] scratch buffer: a third kind of workgroup computes the input term into it, frame by frame.
 *       `membrane` must be NULL.  R of a segment is the same in every layer.
 *   fin[l * n_segs + i]: packed input weights of layer l >= 1; spikes_in must equal segs[(l-1) * n_segs + i].spikes_i8.
 *   rows_per_wg[l]: 4, 8 or 16 (NULL: 8 everywhere).  One workgroup occupies a CU: keep the sum over all layers of
 *       ceil(R / rows_per_wg) within the device's CU count, or layers queue behind each other (correct, not pipelined).
 *   lag: frames a consumer lets its producers run ahead before it (re)starts -- amortises its polls; 16-32 is a good value.
 *   scratch: device memory, sfsn_stack_scratch_bytes(...) bytes, ZEROED by the caller once (before its first use) and private to
 *       the launches that use it, one at a time: every launch leaves its counters zeroed again (its last workgroup does it -- no
 *       memset per launch).  Word 0 is an error flag the caller may read back after a launch: non-zero = a bounded hand-off
 *       wait expired (results invalid); the library never clears it.
 * Shared gate weights only (SFSN_EUNSUPPORTED otherwise: use the per-layer calls). */
size_t sfsn_stack_scratch_bytes(int n_layers, int n_segs, int rows_total);
int sfsn_gsn_stack_scan(const sfsn_scan_segment* segs /* host [n_layers][n_segs] */, const sfsn_fused_input* fin /* host, same shape;
                        layer-0 entries ignored */, int n_layers, int n_segs, int T, int H, const int* rows_per_wg /* host [n_layers] */,
                        int lag, void* scratch, size_t scratch_bytes, void* stream);
/* The same with layer 0's real-valued input product inside the scan for the segments whose fx[i].x is given (round 4; fx NULL or
 * fx[i].x NULL: the segment's layer 0 reads `zin` as above): x [T][R][I] fp32 and W_ih [H][I] fp32 as for
 * sfsn_gsn_layer_scan_fused_x -- the bf16 3-way split of sfsn_input_proj_f32, bit-identical to that call + the scan.  Needs the
 * layout this entry point gives stacks of H <= 224 without input-term buffers (or a single layer): 8 rows per workgroup in every
 * layer, even I <= 64, R a multiple of 8, x 16-byte aligned; SFSN_EUNSUPPORTED otherwise (use `zin`). */
int sfsn_gsn_stack_scan_x(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, const sfsn_fused_x* fx /* host [n_segs], nullable */,
                          int n_layers, int n_segs, int T, int H, const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes,
                          void* stream);
/* Round 6 (ABI 19): the same launch for weights packed with 16 bits (sfsn_w3_pack_bits(w, n, k, 16, ..): the least significant digit plane
 * of every recurrent / spike-input matrix is zero): the zero plane's matrix instructions are skipped in the IO-wave scan, FUSEDX3 and FUSED3
 * roles of the sub-band pair layout (8 rows per workgroup, no input-term buffers for the layers >= 1, H <= 224) -- the same sums, hence the
 * same results as sfsn_gsn_stack_scan_x on such weights.  SFSN_EUNSUPPORTED for every other layout.  BASELINE configs[2]'s 16-bit mode
 * (module.weight_bits = 16): a report mode, not the fp32 parity mode. */
int sfsn_gsn_stack_scan_x_w16(const sfsn_scan_segment* segs, const sfsn_fused_input* fin, const sfsn_fused_x* fx /* host [n_segs], nullable */,
                          int n_layers, int n_segs, int T, int H, const int* rows_per_wg, int lag, void* scratch, size_t scratch_bytes,
                          void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Time-parallel products.
 * sfsn_input_proj_f32: z[m][n] = sum_k x[m][k] * w[n][k] (+ bias[n])  (NEURON:141-142 for layer 0: real-valued x)
 *     fp32-accurate on the bf16 matrix cores: x and w are split (round to nearest) into three bf16 pieces each, the six
 *     leading piece products are exact in fp32 and carry x.w to 2^-26 |x||w| per term, accumulation in fp32
 *     (even K <= 192); v_mfma_f32_16x16x4_f32 (an exact k-ordered fmaf chain) for the other shapes.
 *     x [M][K], w [N][K] row-major fp32, z [M][ldz] (columns 0..N-1 written; ldz > N lets the two gate halves of an
 *     unshared cell land side by side).
 * sfsn_spike_proj:     y[m][n] = dq[n] * sum_k s[m][k] * Wq[n][k] (+ bias[n])
 *     s int8 0/1 [M][pad64(K)] as written by the scan; Wq/dq from sfsn_w3_pack(W [N][K]); y [M][N] fp32.
 *     Used for layer>=1 input products (bias NULL: NEURON:141) and the projection (nn.Linear MODEL:49-52,118;
 *     FROZEN:71-76,125: bias added after the product).
 * ---------------------------------------------------------------------------------------------------- */
int sfsn_input_proj_f32(const float* x, const float* w, const float* bias /* [N], nullable */, float* z, int M, int K, int N,
                        int ldz /* >= N: row stride of z */, void* stream);

/* Several products in ONE launch (ABI 16): the projections (or the layer-0 input products) of the sub-band groups of a chunk are
 * independent of one another (MODEL:118,141 per sequence model); each alone leaves compute units idle and pays a launch boundary.
 * Same results as one sfsn_spike_proj / sfsn_input_proj_f32 call per job (every workgroup runs its job's own tiling).  n <= 8; all
 * spike jobs share ceil(K / 64).  SFSN_EUNSUPPORTED when a job would not take the single entry's fast kernel (N % 4, ld % 4, M < 64,
 * an unaligned output, N > 256, odd K / K > 192 for the real-valued product): issue the calls one by one then. */
typedef struct sfsn_proj_job {
    const int8_t* s;        /* [M][pad64(K)] int8 spikes */
    const int8_t* w_packed; /* sfsn_w3_pack(W [N][K])     */
    const float* w_dq;
    const float* bias;      /* [N], nullable */
    float* y;               /* [M][ldy] */
    int M, K, N, ldy;
} sfsn_proj_job;
int sfsn_spike_proj_multi(const sfsn_proj_job* jobs /* host */, int n, void* stream);
typedef struct sfsn_inproj_job {
    const float* x;    /* [M][K] */
    const float* w;    /* [N][K] */
    const float* bias; /* [N], nullable */
    float* z;          /* [M][ldz] */
    int M, K, N, ldz;
} sfsn_inproj_job;
int sfsn_input_proj_f32_multi(const sfsn_inproj_job* jobs /* host */, int n, void* stream);
int sfsn_spike_proj(const int8_t* s, const int8_t* w_packed, const float* w_dq, const float* bias /* nullable */,
                    float* y, int M, int K, int N, int ldy /* >= N: row stride of y */, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Feature prologue -- replaces MODEL:434-440 (|X|^fdrc, drop Nyquist, band select), SubbandModel.forward's
 * two _freq_unfold calls + cat (MODEL:239-258, 265-312; FROZEN:350-431,451-474), the "b f t -> t b f"
 * transposes (MODEL:108,155) and the input normalisation: nn.LayerNorm (MODEL:27-28,111-112) or the frozen
 * model's offline_laplace_norm (FROZEN:147-169, 475, 578).
 *
 * A feature group g produces x_g [T][B*N][I], row r = b*N + k, I = (ctr + 2 nbr) + (ctr_fb + 2 nbr_fb):
 *     j <  ctr+2nbr : mag[b][reflect(lo + k*ctr - nbr + j)][t]            (reflect: f<0 -> -f, f>nf-1 -> 2(nf-1)-f)
 *     j >= ctr+2nbr : fb[t][b][reflect(lo + k*ctr_fb - nbr_fb + j') % FB]  (the tiled full-band output, MODEL:442-443)
 * with mag = |stft|^fdrc on bins 0..nf-1 (nf = F-1).  The full-band model's own input is the group
 * {lo=0, n_units=1, ctr=FB, nbr=0, ctr_fb=0}.
 * ---------------------------------------------------------------------------------------------------- */
#define SFSN_NORM_NONE 0
#define SFSN_NORM_LAYERNORM 1 /* (x - mean) * rstd * ln_w + ln_b over the I features, eps inside the sqrt       */
#define SFSN_NORM_LAPLACE 2   /* x / (mu[b] + 2.220446049250313e-16), mu from sfsn_laplace_means                 */
#define SFSN_NORM_GAUSSIAN 4   /* (x - mu[b]) / (sd[b] + 2.220446049250313e-16), mu / sd from sfsn_gaussian_stats; sd is passed in ln_w */
#define SFSN_NORM_CUMLAPLACE 3 /* sfsn_stream_hop only: x / (running mean of the row + eps), state carried per row -- the offline
                                  path runs sfsn_features with SFSN_NORM_NONE and then sfsn_cum_laplace_norm             */

typedef struct sfsn_feature_group {
    float* x;           /* out [T][B*n_units][I]                                                             */
    const float* ln_w;  /* [I] (LAYERNORM); GAUSSIAN: [B] the clips' standard deviations                      */
    const float* ln_b;  /* [I] (LAYERNORM)                                                                    */
    const float* mu;    /* [B] (LAPLACE, GAUSSIAN)                                                            */
    int lo, n_units, ctr, nbr, ctr_fb, nbr_fb;
    int norm;           /* SFSN_NORM_*                                                                        */
    float ln_eps;
} sfsn_feature_group;

int sfsn_features(const float* stft_ri /* [B][F][T][2] */, const float* fb_tbf /* [T][B][FB], NULL if unused */,
                  int B, int F, int T, int FB, float fdrc, const sfsn_feature_group* groups /* host */, int n_groups,
                  int t0, int nt /* frames [t0, t0+nt) are produced; tensors are indexed by absolute frame */, void* stream);
/* The same launch with a side job: extra workgroups zero `zero_bytes` bytes at `zero_ptr` (both multiples of 16) -- the zero initial
 * state of the forward's scans (MODEL:100-106) written by the first launch of the forward's chain instead of a fill launch of its own. */
int sfsn_features_z(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc, const sfsn_feature_group* groups,
                    int n_groups, int t0, int nt, float* zero_ptr, size_t zero_bytes, void* stream);

/* The feature prologue AND the layer-0 input product of a chunk in ONE launch (ABI 17) -- replaces, for every job, the pair
 * sfsn_features + sfsn_input_proj_f32 (MODEL:434-440,239-258,108-112 followed by NEURON:141-142): a workgroup keeps the rows it has
 * normalised, splits them into the three bf16 pieces from registers and forms x . W_ih^T + b on the bf16 matrix cores; the rows go
 * to HBM only when `feat.x` is given (the API's all_layer_outputs[0], or a scan that forms its product itself).  A job with w = NULL
 * is sfsn_features alone for that group (feat.x required).  Bit-identical to the two calls: the rows by features_kernel's
 * expressions and wave reductions, the products by input_proj_bf3_kernel's six piece products in its order.
 * z is CHUNK-LOCAL: [nt][B * n_units][ldz], frame t0 first (x is indexed by absolute frame, as in sfsn_features).
 * SFSN_EUNSUPPORTED for what sfsn_input_proj_f32 would not run on its bf16-split kernel (odd I, I > 160 with H > 128, H > 384,
 * fewer than 64 rows, SFSN_INPROJ_F32 set) and for groups reading more than 144 bins or FB > 128: issue the two calls then. */
typedef struct sfsn_featproj_job {
    sfsn_feature_group feat; /* feat.x nullable when w is given */
    const float* w;          /* [H][I] fp32 W_ih (shared gates), NULL = rows only                                */
    const float* bias;       /* [H], nullable                                                                     */
    float* z;                /* out [nt][B*n_units][ldz]                                                          */
    int H, ldz;
} sfsn_featproj_job;
int sfsn_features_proj(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                       const sfsn_featproj_job* jobs /* host */, int n_jobs, int t0, int nt, float* zero_ptr /* as sfsn_features_z */,
                       size_t zero_bytes, void* stream);

/* Per-clip means for offline_laplace_norm (FROZEN:162-164: mean over all non-batch dims of the gathered,
 * un-normalised group tensor).  mu_out [n_groups][B].  Two launches: row sums of mag / fb, then the
 * weighted combination; `scratch` needs B*(F-1+FB) floats. */
int sfsn_laplace_means(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                       const sfsn_feature_group* groups /* host, only geometry fields are read */, int n_groups,
                       float* mu_out, float* scratch, void* stream);
/* Per-clip mean and UNBIASED standard deviation for offline_gaussian_norm (FROZEN:205-218: torch.mean / torch.std over all non-batch
 * dims of the gathered, un-normalised group tensor).  mu_out, sd_out [n_groups][B].  `scratch`: 5 * B * (F - 1 + FB) + 2 floats,
 * 8-byte aligned.  T >= 2. */
int sfsn_gaussian_stats(const float* stft_ri, const float* fb_tbf, int B, int F, int T, int FB, float fdrc,
                        const sfsn_feature_group* groups /* host, only geometry fields are read */, int n_groups, float* mu_out,
                        float* sd_out, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Deep-filter epilogue -- replaces the output re-index of SubBandSequenceModel.forward (MODEL:160-167,
 * FROZEN:259-265), deepfiltering (MODEL:315-346, FROZEN:15-39) and the reconstruction MODEL:450-472 /
 * FROZEN:588-607 (cat groups, Nyquist bin passes through, |.|).
 *     Y[b][s][f][t] = sum_{d<df} X[b][f][t-(df-1)+d] * C[d]   (zero for t-(df-1)+d < 0)
 *     C[d] = proj[t][b*N+k][((0*fc+fci)*df+d)*S+s] + i proj[t][b*N+k][((1*fc+fci)*df+d)*S+s],  f = lo + k*fc + fci
 * Groups are laid end to end from bin 0; bins not covered (>= sum N*fc, at least Nyquist) are copied.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct sfsn_df_group {
    const float* proj; /* [T][B*n_units][2*fc*df*S] */
    int n_units, fc, df;
} sfsn_df_group;

int sfsn_deepfilter(const float* stft_ri /* [B][F][T][2] */, int B, int F, int T, int S,
                    const sfsn_df_group* groups /* host */, int n_groups, float* enh_ri /* [B][S][F][T][2] */,
                    float* enh_mag /* [B][S][F][T], nullable */, int t0, int nt /* frames [t0, t0+nt) */, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Round 6: the sub-band epilogue in ONE launch -- per group the projection of the last layer's spikes (MODEL:118 `self.proj`,
 * FROZEN:125 `fc_output_layer`: sfsn_spike_proj's product) AND everything sfsn_deepfilter does with its result (MODEL:160-167,
 * 315-346, 450-474).  The coefficient tile of a (clip, 16-frame block, unit range) is formed on the int8 matrix cores, stays in LDS,
 * is written to `proj` once (the module API's last all_layer_outputs entry; NULL: not written at all) and is applied to the noisy
 * spectrum from LDS: the coefficients are never re-read from memory.  Same results, bit for bit, as sfsn_spike_proj_multi followed
 * by sfsn_deepfilter on the same arguments.  H <= 256 (spike rows of pad64(H) bytes), every P = 2*fc*df*S a multiple of 4 and <= 256;
 * SFSN_EUNSUPPORTED otherwise (issue the two calls).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct sfsn_projdf_group {
    const int8_t* spikes_i8; /* [T][B*n_units][pad64(H)] int8 spikes of the group's last layer (frame 0 of the sequence) */
    const int8_t* w_packed;  /* sfsn_w3_pack(W_p [P][H]) */
    const float* w_dq;
    const float* bias;       /* [P], nullable */
    float* proj;             /* [T][B*n_units][P] coefficient rows, 16-byte aligned; NULL = not written */
    int n_units, fc, df;
} sfsn_projdf_group;
int sfsn_proj_deepfilter(const float* stft_ri /* [B][F][T][2] */, int B, int F, int T, int S, int H,
                         const sfsn_projdf_group* groups /* host */, int n_groups, float* enh_ri /* [B][S][F][T][2] */,
                         float* enh_mag /* [B][S][F][T], nullable */, int t0, int nt /* frames [t0, t0+nt) */, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Streaming sessions (BASELINE configs[4]): the input history the deep filter reaches back into (MODEL:331-333: df - 1 frames
 * of the noisy spectrum).  hist [rows][D + hop] complex64, rows = B * F: frames [hop, hop + D) move to [0, D) and the `hop`
 * new frames of inp [rows][hop] are appended, in place, in one launch.  D + hop <= 16 (SFSN_EUNSUPPORTED beyond).
 * ---------------------------------------------------------------------------------------------------- */
int sfsn_hist_shift(float* hist_ri, const float* inp_ri, int rows, int D, int hop, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * cumulative_laplace_norm (FROZEN:172-202 in the form that accepts the 5-D sub-band tensor,
 * recipes/intel_ndns/spiking_fullsubnet_freeze_phase/model_low_freq_count_time.py:182-204; FROZEN's own version raises on
 * the sub-band input): every row (clip x unit) of x [T][R][I] is divided, frame by frame, by the mean of everything the row
 * has seen so far: mean[r][t] = sum_{t' <= t} sum_i x[t'][r][i] / (I * (frames_before + t + 1)), x /= mean + 2.22e-16.
 * In place.  cum_state [R] (nullable): the running sums, in/out -- a caller that feeds a sequence in pieces passes the same
 * buffer and the number of frames already seen.  scratch: T * R floats.
 * ---------------------------------------------------------------------------------------------------- */
int sfsn_cum_laplace_norm(float* x /* [T][R][I] */, int T, int R, int I, float* cum_state /* [R], nullable */,
                          int frames_before, float* scratch /* [T][R] */, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Streaming hop -- BASELINE configs[4]: `hop` new frames of B clips through the WHOLE live model (features, every GSN layer
 * of the full-band model and of the sub-band groups, projections, deep filter: MODEL:429-473 restricted to the new frames)
 * in ONE launch, with the (h, c) state of every layer (NEURON:50-62) and the deep filter's input history carried in device
 * memory between launches.  Replaces the ~15 launches the offline kernels need per hop (each costs ~4.5 us of launch
 * boundary on this hardware, which is all a one-frame hop consists of).
 *
 * One wave owns one 16-neuron tile of one layer for 16 rows and keeps that tile's weights in registers; the workgroups of
 * consecutive stages (full-band layer 0 -> ... -> sub-band layer 0, which also computes the full-band projection it needs
 * -> ... -> sub-band projection + deep filter) hand each frame over through L2 as data-tagged granules (a write-through
 * store carries the spikes AND the launch's tag; consumers poll the payload itself).
 * The recurrent product of a layer is issued before its input has arrived (it needs the previous frame only).
 * Arithmetic is that of sfsn_features / sfsn_spike_proj / sfsn_gsn_layer_scan / sfsn_deepfilter expression by expression;
 * the real-valued input product is sfsn_input_proj_f32's fp32-MFMA form with four accumulators.
 *
 * Shared or separate gate weights (sfsn_hop_desc.unshared), LayerNorm / cumulative Laplace / no normalisation, H % 16 == 0, H <= 320, I <= 192, P <= 256 (full-band P <= 128), at most
 * SFSN_HOP_MAX_LAYERS layers and SFSN_HOP_MAX_GROUPS groups, D + hop <= 32, and few enough rows that every wave tile gets
 * its own compute unit (SFSN_EUNSUPPORTED otherwise: the caller then runs the per-kernel sequence).
 * ---------------------------------------------------------------------------------------------------- */
#define SFSN_HOP_MAX_LAYERS 3
#define SFSN_HOP_MAX_GROUPS 4
typedef struct sfsn_hop_layer {
    /* (desc.unshared: every image below covers the 2H rows of both gates, the forget gate's H rows first: [2H/16] fragment tiles,
     *  sfsn_w3_pack(W [2H][.]) and dq vectors of 2H entries) */
    const float* w_ih_frag; /* layer 0: fp32 W_ih [H][I] in MFMA fragment order [H/16][KC][64][4], KC = ceil(I/16):
                               element ((tile*KC + c)*64 + 16*q + n)*4 + e = W_ih[16*tile + n][16*c + 4*q + e], zero where
                               the column index is >= I (NULL for layers >= 1)                                      */
    const int8_t* w_ih;     /* layers >= 1: sfsn_w3_pack(W_ih [H][H]) (NULL for layer 0)                            */
    const float* w_ih_dq;
    const int8_t* w_hh;     /* sfsn_w3_pack(W_hh [H][H])                                                            */
    const float* w_hh_dq;
    const float* bias;      /* [2H] bias_ih                                                                         */
    const float* bn_alpha;  /* [H]                                                                                  */
    const float* bn_beta;   /* [H]                                                                                  */
    int8_t* h[2];           /* [R][pad64(H)] x 2: spikes of the last frame, in/out; launch k (= launch_index) reads
                               h[k & 1] and writes h[(k + 1) & 1]; zero both to reset                                 */
    float* c;               /* [R][H] membrane, in/out                                                              */
    int8_t* spikes;         /* [hop][R][pad64(H)] scratch, zeroed by the caller once (bit 0 of a byte = the spike,
                               bits 1..7 = the tag of the launch that wrote it)                                      */
} sfsn_hop_layer;
typedef struct sfsn_hop_seq {      /* one sequence model: the full-band model or one sub-band group                  */
    sfsn_hop_layer layer[SFSN_HOP_MAX_LAYERS];
    int n_layers, H, P;
    sfsn_feature_group feat;       /* geometry + normalisation of this model's input rows (`x` and `mu` unused);
                                      rows R = B * feat.n_units, row b * n_units + k                                 */
    const int8_t* w_p;             /* sfsn_w3_pack(proj.weight [P][H])                                               */
    const float* w_p_dq;
    const float* b_p;              /* [P]                                                                            */
    int df;                        /* deep-filter order of a sub-band group; 0 for the full-band model               */
    int fc;                        /* sub-band group: centre bins per unit (P == 2 * fc * df * S)                    */
    float* cum[2];                 /* feat.norm == SFSN_NORM_CUMLAPLACE: [R] x 2, the rows' running sums, in/out (launch k reads
                                      cum[k & 1], writes cum[(k + 1) & 1]; zero both to reset); NULL otherwise          */
} sfsn_hop_seq;
typedef struct sfsn_hop_desc {
    sfsn_hop_seq fb;
    sfsn_hop_seq sb[SFSN_HOP_MAX_GROUPS];
    int n_groups;
    int B, F, S, hop, D;           /* D = max(df) - 1 frames of history                                              */
    float fdrc;
    const float* inp_ri;           /* [B][F][hop][2] the new noisy frames                                            */
    float* hist_ri;                /* [B][F][D][2]   the last D noisy frames, in/out (zero to reset); NULL if D == 0 */
    float* enh_ri;                 /* [B][S][F][hop][2] out                                                          */
    float* enh_mag;                /* [B][S][F][hop] out, nullable                                                   */
    void* scratch;                 /* sfsn_hop_scratch_bytes(desc) bytes, ZEROED by the caller once; word 0 = error flag
                                      (non-zero after a launch: a bounded hand-off wait expired, results invalid)     */
    size_t scratch_bytes;
    unsigned launch_index;         /* launches made on this state so far: the caller adds 1 after every launch (it picks
                                      the hand-off tag and the parity of the double-buffered state; a launch must not
                                      be replayed with the index of its predecessor, so no graph capture)            */
    /* Waveform mode (hop == 1, n_fft 512 / hop_length 128; wave_in != NULL): the launch also computes the new frame's
     * spectrum from the samples (torch.stft(center=True) framing: the frame that ends with these 128 samples, i.e. frame
     * frame_index of the utterance when the caller has fed 128 * (frame_index + 2) samples) and turns the enhanced frame back
     * into samples (torch.istft's overlap-add and envelope): wave_out receives the 128 padded positions
     * [128 * frame_index, 128 * frame_index + 128), i.e. output samples [128 * (frame_index - 2), 128 * (frame_index - 1)) --
     * the first two launches of an utterance write the part torch.istft trims.  inp_ri is ignored. */
    const float* wave_in;          /* [B][128] the new samples                                                       */
    float* wave_state;             /* [B][512] the last 512 input samples, in/out (zero = start of an utterance)      */
    float* ola_state;              /* [B][S][512] overlap-add accumulator of the output, in/out (zero to reset)       */
    float* wave_out;               /* [B][S][128] out                                                                */
    const float* window;           /* [512] analysis / synthesis window (the reference: hann)                       */
    float* spec_g;                 /* [B][F][4] scratch ({re, tag, im, tag} granules of the noisy frame), zeroed once */
    float* enh_g;                  /* [B][S][F][4] scratch (granules of the enhanced frame), zeroed once              */
    int frame_index;
    unsigned* done;                /* nullable, [B][S]: word (b, s) is set to launch_index + 1 (system-scope release) once wave_out
                                      (b, s) has been written.  wave_in, wave_out and done may be pinned host memory the device can
                                      reach: samples in host memory -> enhanced samples in host memory with no copy launch and no
                                      stream synchronisation (the caller spins on the words)                               */
    int frames_before;             /* frames this state has seen since it was zeroed (the caller adds `hop` after every launch):
                                      SFSN_NORM_CUMLAPLACE's denominator                                             */
    int unshared;                  /* non-zero: separate forget / cell gate weights (shared_weights = False, NEURON:137-139): every
                                      layer's w_hh / w_ih images and dq vectors cover 2H rows, forget rows first (ABI 15)      */
} sfsn_hop_desc;

size_t sfsn_hop_scratch_bytes(const sfsn_hop_desc* desc /* host */);
int sfsn_stream_hop(const sfsn_hop_desc* desc /* host */, void* stream);

/* The RESIDENT form of the waveform hop (BASELINE.json configs[4] names a "persistent kernel"; round 3).  One launch serves hop
 * after hop of a waveform session with host completion words (wave_in, wave_out and done in pinned host memory, hop == 1):
 * for hop k = 0, 1, ... the host writes the 128 samples per clip into wave_in, then stores k + 1 into the 32-bit `doorbell`
 * word (pinned host memory too), then waits until every done word reads desc->launch_index + k + 1 -- only then may it ring
 * hop k + 1.  The kernel uses launch index desc->launch_index + k, frame index desc->frame_index + k and
 * desc->frames_before + k for hop k (what k separate sfsn_stream_hop calls would have been given).  Storing 0xFFFFFFFF ends the
 * kernel; so does a doorbell silent for about `idle_ms` milliseconds and an expired hand-off wait (error word set, as for a
 * launch): the kernel never outlives its caller by more than that.  The caller keeps `stream` free of other work while
 * the kernel is resident (it holds the hop's workgroups -- 23 of 256 CUs for the M model at B = 1) and adds the hops served to
 * its own launch / frame counters afterwards; doorbell[1] (zeroed by the caller before the call) is set to 1 by the kernel when it
 * leaves.  Same results as the launches, bit for bit. */
int sfsn_stream_hop_resident(const sfsn_hop_desc* desc /* host */, void* doorbell /* pinned host, u32[2] */, unsigned idle_ms,
                             void* stream);
/* Diagnostic: the launch's stages in block order, out[4 * i] = {sequence (0 = full-band, 1 + g = group g), layer (-1 = projection
 * [+ deep filter]), first workgroup, workgroups}; returns the number of stages (or a negative status).  With SFSN_HOP_DEBUG set
 * in the environment a launch leaves eight 100 MHz time stamps per wave (4 waves per workgroup) in `scratch`, behind the
 * 64 bytes of control words: entry, set-up done, recurrent half issued, full-band projection arrived (sub-band layer 0),
 * input arrived, frame computed, deep filter done, exit (scripts/exp_hop.py prints them per stage). */
int sfsn_hop_stages(const sfsn_hop_desc* desc /* host */, int* out /* host [cap][4] */, int cap);

/* ----------------------------------------------------------------------------------------------------
 * Spike counts -- the only thing the reference's energy proxy reads from the spike tensors:
 * compute_synops (audiozen/metric.py:303-327) uses torch.gt(layer_output, 0).float().mean() per layer.  Counting the
 * int8 spikes the scan already writes lets a caller that only wants SynOPs skip the fp32 spike tensors
 * (spikes_f32 = NULL in sfsn_gsn_layer_scan): 4 B/spike of HBM writes and a 3-pass torch reduction less.
 *     *count += #{ spikes_i8[i] != 0, i < n_bytes }       (exact integer; pad columns of the int8 layout are zero)
 * One launch for up to SFSN_MAX_COUNT_TENSORS tensors.  `count` is accumulated (the caller zeroes it).
 * ---------------------------------------------------------------------------------------------------- */
#define SFSN_MAX_COUNT_TENSORS 16
typedef struct sfsn_count_tensor {
    const int8_t* spikes_i8;     /* device, 16-byte aligned, n_bytes % 16 == 0 */
    unsigned long long n_bytes;
    unsigned long long* count;   /* device */
} sfsn_count_tensor;

int sfsn_spike_count(const sfsn_count_tensor* tensors /* host */, int n_tensors, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * The two edges of the path -- replace audiozen/acoustics/audio_feature.py:236-347 as the models call them
 * (MODEL:429,473; FROZEN:568-572,612-617): torch.stft(y, n_fft, hop, n_fft, window, center=True, pad_mode="constant",
 * return_complex=True) and torch.istft(X, n_fft, hop, n_fft, window, length=length).  n_fft = 512 (every reference config;
 * SFSN_EUNSUPPORTED otherwise), n_fft % hop == 0, 4 hops per window at most.  `window` is the analysis/synthesis window on
 * the device (n_fft floats; the reference passes torch.hann_window(n_fft)).
 *   stft : frame t = samples [t*hop - n_fft/2, t*hop + n_fft/2), zero outside [0, L);  T must be 1 + L / hop
 *   istft: y[m] = sum_t w[n - t*hop] * irfft(X[:, t])[n - t*hop] / sum_t w[n - t*hop]^2,  n = m + n_fft/2,  m < length
 *          (imaginary parts of the DC and Nyquist bins are ignored, as by a complex-to-real transform)
 * ---------------------------------------------------------------------------------------------------- */
int sfsn_stft(const float* wave /* [B][L] */, int B, int L, int n_fft, int hop, const float* window,
              float* stft_ri /* [B][n_fft/2+1][T][2] */, int T, void* stream);
int sfsn_istft(const float* stft_ri /* [B][n_fft/2+1][T][2] */, int B, int T, int n_fft, int hop, const float* window,
               float* wave /* [B][length] */, int length, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SFSN_H */
