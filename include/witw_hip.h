/* libwitw_hip.so — C ABI of the MI355X (gfx950) hot path of IQTLabs/WITW.
 *
 * The reference exposes no FFI: its boundary is the Python surface of model/cvig_fov.py
 * (SURVEY.md §8b). Each entry point below replaces the stock-PyTorch op(s) behind one of those
 * Python functions; witw_amd/_lib.py binds them with ctypes and witw_amd/cvig_fov.py keeps the
 * reference's names on top (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller unless a
 *    comment says HOST; the library never allocates or frees;
 *  - fp32 everywhere (indices: int64 orientation, int32 ranks); tensors are contiguous;
 *  - every launcher is asynchronous on `stream` (a hipStream_t passed as void*, NULL = default
 *    stream), never synchronises, is re-entrant across streams;
 *  - return 0 on success, <0 on error (-1 invalid argument, -2 launch failure, -3 no gfx950
 *    device); the message is thread-local text from witw_last_error();
 *  - one process per GPU; the code object is gfx950 only.
 */
#ifndef WITW_HIP_H
#define WITW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

const char* witw_last_error(void);
int witw_version(void);               /* major*10000 + minor*100 + patch */
int witw_device_check(int device);    /* 0 iff `device` is a gfx950 part */
/* name of the conv kernel instantiation the calling thread's last conv launcher picked, spelled as in rocprof kernel names
 * (e.g. "conv3x3_bf16_s16_kernel<false>"); "" before the first launch. Parity tests assert on it so that a change of the
 * launcher's thresholds cannot silently move a test onto another kernel. */
const char* witw_last_kernel_variant(void);

/* ---- FOV_DSM encoder: Conv2d(3x3,pad 1) [+HorizCircPadding] [+Dropout2d] [+ReLU] [+MaxPool2d(2)]
 *      reference: model/cvig_fov.py:212-231 (padding), :234-245 (dropout), :256-294 (layer stack).
 * Activations are NHWC with the channel count padded to a multiple of 8. */

/* output-channel tile the kernel will use for `cout` (64 or 128) */
int witw_conv3x3_tile_n(int cout);
/* waves per workgroup (4 or 8) chosen for a layer shape; with tile_n, stride and pool it names the kernel
 * instantiation conv3x3_nhwc_f32_kernel<tile_n,stride_h,pool,waves> seen in profiles */
int witw_conv3x3_workgroup_waves(int B, int H, int W, int Cout, int stride_h);
/* number of floats of the packed filter / of the zero-padded bias for a (cout, cin) layer */
long long witw_conv3x3_packed_floats(int cout, int cin);
int witw_conv3x3_bias_floats(int cout);
/* w_kcrs: torch layout [cout][cin][3][3]. transpose_flip=1 packs the data-gradient filter instead
 * (source layout [cin][cout][3][3], taps rotated 180 degrees). */
int witw_conv3x3_pack_weights(const float* w_kcrs, float* wpk, int cout, int cin, int transpose_flip, void* stream);
/* NCHW [B,C,H,W] (C<=8) -> NHWC8 [B,H,W,8], zero-filled channels; replaces the implicit layout of
 * the first Conv2d call (model/cvig_fov.py:293). */
int witw_nchw_to_nhwc8(const float* x, float* y, int B, int C, int H, int W, void* stream);
/* x [B,H,W,Cin] (Cin%8==0) -> y NHWC [B,Hy,Wy,Cout] (or NCHW [B,Cout,Hy,Wy] if out_nchw).
 * stride_h in {1,2} (stride_w = 1); pad_circular: wrap columns (HorizCircPadding) else zero pad;
 * dropmask: NULL or [B,Cout] Dropout2d scales applied before the ReLU; pool: fused MaxPool2d(2,2).
 * bias: witw_conv3x3_bias_floats(Cout) floats. */
int witw_conv3x3_fwd(const float* x, const float* wpk, const float* bias, const float* dropmask, float* y, int B, int H,
                     int W, int Cin, int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw,
                     void* stream);

/* First layer fast path: Conv2d(C<=4 -> 64, 3x3, pad 1)+ReLU straight from the NCHW fp32 image (features[0..1],
 * model/cvig_fov.py:256-260): wf = witw_conv3x3_first_pack(w [64][C][3][3]) (a 2560-float buffer; round_bf16 = 1
 * writes the bf16 MFMA kernel's filter image instead and goes with out_bf16 = 1), y = NHWC [B,H,W,64] fp32, or bf16
 * with bf16-rounded operands on v_mfma_f32_32x32x16_bf16 when out_bf16 = 1 (this form takes C <= 8: cvig_semantic's 5-channel
 * first conv, model/cvig_semantic.py:301-303, at inference); out_bf16 = 2: fp32 arithmetic, split-fp16 output
 * [B,H,W,8,2,8] (hi / lo planes per 8 channels) for the fp16x3 path. */
int witw_conv3x3_first_pack(const float* w, float* wf, int C, int round_bf16, void* stream);
/* Layers 0 and 2 of the encoder fused, bf16 inference: x NCHW fp32 [B,C<=8,H,W] -> y NHWC bf16 [B,H/2,W/2,64] =
 * MaxPool2d(2)(ReLU(conv 64->64 (ReLU(conv C->64 (x))))) (features[0..4], model/cvig_fov.py:256-260), the 64-channel map between
 * the two convs never leaving the chip. wf0 / bias0: witw_conv3x3_first_pack(round_bf16 = 1) image and bias; wpk2 / bias2:
 * witw_conv3x3_bf16_pack_weights image (64 -> 64) and bias. Bit-identical to witw_conv3x3_first_fwd(out_bf16 = 1) followed by
 * witw_conv3x3_bf16_fwd(relu, pool). */
int witw_conv_first2_bf16_fwd(const float* x, const void* wf0, const float* bias0, const void* wpk2, const float* bias2, void* y,
                              int B, int C, int H, int W, int pad_circular, void* stream);
/* The training form (cvig_semantic trains layer 0, model/cvig_semantic.py:301-309, so the backward crosses both layers): the same
 * y, plus pool_code [B,H/2,W/2,64] (arg-max of each 2x2 window, dy*2+dx: the input of witw_maxpool2x2_bwd_bf16) and gate_bits
 * [B,H,W,8] bytes (bit c & 7 of byte c >> 3 = layer 0's output channel c at that pixel is > 0) instead of the two 64-channel
 * activations. Layer 2's own ReLU gate is y > 0. H and W even. */
int witw_conv_first2_bf16_fwd_train(const float* x, const void* wf0, const float* bias0, const void* wpk2, const float* bias2, void* y,
                                    unsigned char* pool_code, unsigned char* gate_bits, int B, int C, int H, int W, int pad_circular,
                                    void* stream);
/* Data gradient of a 64-input-channel stride-1 layer whose upstream ReLU gate is given as ONE BIT per output (gate_bits
 * [B,H,W,Cout/8] bytes as above) instead of a bf16 tensor: same bits as witw_conv3x3_bf16_fwd_ex(gate = the activation). Runs on the
 * weight-resident kernel only: witw_conv3x3_bf16_gatebits_ok tells whether a shape qualifies (1) or the tensor gate has to be kept (0). */
int witw_conv3x3_bf16_gatebits_ok(int B, int H, int W, int Cin, int Cout);
int witw_conv3x3_bf16_fwd_gatebits(const void* x_bf16, const void* wpk_bf16, const float* bias, const void* gate_bits, void* y, int B, int H,
                                   int W, int Cin, int Cout, int pad_circular, int relu, void* stream);
int witw_conv3x3_first_fwd(const float* x, const float* wf, const float* bias, void* y, int B, int C, int H, int W,
                           int pad_circular, int relu, int out_bf16, void* stream);

/* Extended form used by the backward pass: `gate` (NULL or a tensor shaped like y) zeroes outputs where
 * gate <= 0 (ReLU backward), dilate_h=1 reads the input as zero-interleaved rows (row 2i = physical row i,
 * H is the dilated height) — the data gradient of a stride-(2,1) conv is then this same kernel on the
 * transpose_flip filter (autograd of torch.nn.Conv2d at model/cvig_fov.py:460). relu: 0 none, 1 ReLU,
 * 2 LeakyReLU(lrelu_slope); post_scale/post_shift (NULL or [Cout]): per-channel affine after the activation
 * (eval-mode BatchNorm2d of model/cvig_baseline.py:267-275). */
int witw_conv3x3_fwd_ex(const float* x, const float* wpk, const float* bias, const float* dropmask, const float* gate,
                        const float* post_scale, const float* post_shift, float* y, unsigned char* pool_code, int B, int H,
                        int W, int Cin, int Cout, int stride_h, int pad_circular, int relu, float lrelu_slope, int pool,
                        int out_nchw, int dilate_h, void* stream);
/* 1 (default): a zero-interleaved launch of witw_conv3x3_fwd_ex (dilate_h: the data gradient of a stride-(2,1) conv, model/cvig_fov.py:
 * 263-272 through autograd) on the 8-wave 128-channel tile skips the products whose input rows are the interleaved zeros -- half the
 * MFMAs, the same bits; 0: all products are issued. enable < 0 only queries; returns the previous setting. */
int witw_conv3x3_dil_skip(int enable);
/* backward of the fused MaxPool2d(2,2): pool_code (written by the forward when non-NULL: first arg-max position
 * dy*2+dx in torch's scan order) routes dy [B,Hp,Wp,C] into dx [B,H,W,C] (H>=2Hp, W>=2Wp). */
int witw_maxpool2x2_bwd(const float* dy, const unsigned char* pool_code, float* dx, int B, int Hp, int Wp, int H, int W,
                        int C, void* stream);
/* NCHW [B,C,H,W] -> NHWC [B,H,W,Cpad] with zero-filled extra channels (embedding gradients). */
int witw_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, int Cpad, void* stream);
/* Weight / bias gradient of one conv layer. x: the layer's NHWC input [B,H,W,Cin], dz: gradient at its output
 * [B,Ho,W,Cout] NHWC (ReLU / dropout gates already applied). dw: torch layout [Cout][cin_real][3][3];
 * db: [Cout] or NULL; accumulate != 0 adds into dw/db. workspace: witw_conv3x3_wgrad_workspace_floats(). */
int witw_conv3x3_wgrad_splits(int B, int Ho, int Wo, int Cin, int Cout);
long long witw_conv3x3_wgrad_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h);
int witw_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* db, float* workspace, int B, int H, int W, int Cin,
                       int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, void* stream);
/* torch.optim.Adam step (no weight decay / amsgrad), optimizer of model/cvig_fov.py:416-418. step counts from 1. */
int witw_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                   float beta2, float eps, int step, void* stream);
/* The same step for `count` parameter tensors in ceil(count / 48) launches: HOST arrays of device pointers, element counts and
 * 1-based step numbers (the loop over optimizer.param_groups of torch.optim.Adam.step, model/cvig_fov.py:463). */
int witw_adam_step_multi(float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                         const long long* n, const int* step, int count, float lr, float beta1, float beta2, float eps, void* stream);

/* ---- matching: correlation (:297-315) + crop_overhead (:318-343) + l2_distance (:346-363) fused.
 * ov [Bo,16,4,64], su [Bs,16,4,We] (NCHW embeddings). Outputs [Bo,Bs]; any of them may be NULL.
 * workspace: witw_match_workspace_floats(Bo,Bs) floats (window norms + surface norms). */
long long witw_match_workspace_floats(int Bo, int Bs);
int witw_match_fwd(const float* ov, const float* su, int Bo, int Bs, int We, long long* orientation, float* distance,
                   float* score, float* workspace, void* stream);
/* The same match through the row spectra (retrieval, BASELINE config 5: every gallery row against every query). The orientation
 * search is a circular cross-correlation along the 64 columns: with the 64-point DFT of every (channel,row) line of both sides it
 * costs 21k FLOP per pair instead of 524k. witw_match_spectrum: emb [B,64 lines,W] (overhead: W = 64; surface: W = We, zero-
 * padded) -> spec [B,32,128] (witw_match_spectrum_floats(B) floats; fp64 transform rounded once to fp32; the order of the two
 * 8-byte chunks inside each 16-byte slot depends on the side -- `role` -- and, for overheads, on bit 4 of the embedding's index, so
 * that the match kernel's LDS operand reads are free of bank conflicts: a spectrum is only meaningful to witw_match_fwd_dft, on the
 * side and at the row numbering (mod 32) it was computed for). witw_match_fwd_dft:
 * outputs as witw_match_fwd (first maximum wins); scores agree with the direct sum to fp32 rounding (1e-6 of |ov||su|), so
 * orientations can differ from witw_match_fwd only between shifts whose scores tie to that accuracy.
 * workspace: witw_match_dft_workspace_floats(Bo,Bs) floats. */
long long witw_match_spectrum_floats(long long n_embeddings);
int witw_match_spectrum(const float* emb, float* spec, int B, int W, int role /* 0 surface (query) side, 1 overhead (gallery) side */,
                        void* stream);
long long witw_match_dft_workspace_floats(int Bo, int Bs);
int witw_match_fwd_dft(const float* ov, const float* su, const float* spec_ov, const float* spec_su, int Bo, int Bs, int We,
                       long long* orientation, float* distance, float* score, float* workspace, void* stream);
/* the same, also writing gap [Bo,Bs] = best score - runner-up score over the 64 shifts (distance of the arg-max from a tie) */
int witw_match_fwd_dft_gap(const float* ov, const float* su, const float* spec_ov, const float* spec_su, int Bo, int Bs, int We,
                           long long* orientation, float* distance, float* score, float* gap, float* workspace, void* stream);
/* backward of witw_match_fwd (orientation is a constant of the graph): grad_distance [Bo,Bs] ->
 * grad_ov [Bo,16,4,64] and/or grad_su [Bs,16,4,We]; orientation/score/workspace as left by the forward. scratch: NULL,
 * or witw_match_bwd_scratch_floats(Bo,Bs,We) floats that let the surface gradient split the overheads over several
 * workgroups per surface (fixed-order sum; a batch of 128 surfaces alone leaves half the chip idle). */
long long witw_match_bwd_scratch_floats(int Bo, int Bs, int We);
int witw_match_bwd(const float* ov, const float* su, const long long* orientation, const float* score, const float* workspace,
                   const float* grad_distance, float* grad_ov, float* grad_su, float* scratch, int Bo, int Bs, int We,
                   void* stream);
/* compatibility entries with the reference's materialising semantics */
int witw_crop_overhead(const float* ov, const long long* orientation, float* out /*[Bo,Bs,16,4,We]*/, int Bo, int Bs,
                       int We, void* stream);
int witw_l2_distance(const float* cropped /*[Bo,Bs,n]*/, const float* su /*[Bs,n]*/, float* distance, int Bo, int Bs, int n,
                     void* stream);
/* ranks[q] = #{o : D[o][q] <= D[q+true_offset][q]} — the loop body of test(), model/cvig_fov.py:550-552 */
int witw_rank_count(const float* distance /*[Bo,Bs]*/, int* ranks /*[Bs]*/, int Bo, int Bs, int true_offset, void* stream);

/* k smallest distances of every query column of D[Bo][Bs], ordered by (distance, gallery index): values [Bs][k],
 * indices [Bs][k] (+row_offset: global row number of this shard's first row; -1 pads when Bo < k). 1 <= k <= 32. */
int witw_topk_smallest(const float* distance, float* values, long long* indices, int Bo, int Bs, int k, long long row_offset,
                       void* stream);
/* the same with the gallery rows split over workgroups (one workgroup column per 64 queries cannot fill the chip when the
 * gallery is long): workspace = witw_topk_workspace_bytes(Bo,Bs,k) bytes (NULL or 0 bytes: single pass). Same result. */
long long witw_topk_workspace_bytes(int Bo, int Bs, int k);
int witw_topk_smallest_ws(const float* distance, float* values, long long* indices, int Bo, int Bs, int k, long long row_offset,
                          void* workspace, void* stream);
/* sharded-gallery form: ranks[q] = #{o in this shard : D[o][q] <= threshold[q]} (threshold = true match's distance) */
int witw_rank_count_thresh(const float* distance, const float* threshold, int* ranks, int Bo, int Bs, void* stream);

/* Index-exact retrieval on top of the spectral pass (model/cvig_fov.py:547-552 ranks / top-k must not depend on which of two
 * equally accurate fp32 summation orders produced a distance). witw_match_pairs: a LIST of n_pairs (overhead row pair_o[i],
 * surface row pair_s[i]) pairs -> orientation / distance / score [n_pairs] (any may be NULL), bit-identical to the entries
 * witw_match_fwd writes for those pairs (same MFMA accumulation chain, one wave per pair); wn [Bo,64] / sn [Bs] = the norms
 * at the start of a witw_match_fwd / witw_match_fwd_dft workspace over the same ov / su (window norms, then surface norms).
 * witw_rank_count_band: D [Bo,Bs] known to +-eps: counts[q] = #{o : D[o][q] < threshold[q] - eps}; the (o,q) inside the band
 * |D - threshold| <= eps are appended to pair_o / pair_s (up to `capacity`), *n_pairs = how many there were in all. */
int witw_match_pairs(const float* ov, const float* su, const float* wn, const float* sn, const int* pair_o, const int* pair_s,
                     int n_pairs, int Bo, int Bs, int We, long long* orientation, float* distance, float* score, void* stream);
int witw_rank_count_band(const float* distance, const float* threshold, float eps, int* counts, int* pair_o, int* pair_s,
                         int* n_pairs, int capacity, int Bo, int Bs, void* stream);
/* witw_rank_count_band's list resolved without reading its length back to the host (model/cvig_fov.py:547-552: the rank is a
 * count of distances <= the true match's): re-scores the first min(*n_pairs_dev, capacity) pairs as witw_match_pairs does and
 * adds 1 to counts[pair_s[i]] for every pair with exact distance <= threshold[pair_s[i]]. */
int witw_match_pairs_count(const float* ov, const float* su, const float* wn, const float* sn, const int* pair_o, const int* pair_s,
                           const int* n_pairs_dev, int capacity, int Bo, int Bs, int We, const float* threshold, int* counts, void* stream);
/* which kernel witw_match_pairs runs: 1 = v_fma_f32 chain on the vector pipe, one lane per shift (default, round 6); 0 = the
 * v_mfma_f32_32x32x2_f32 chain of rounds 4-5. Same bits (the f32 MFMA accumulates as a fused-multiply-add chain in k order).
 * impl < 0 only queries; returns the previous setting. */
int witw_match_pairs_impl(int impl);

/* ---- Dropout2d masks (AddDropout, model/cvig_fov.py:234-245): out [n_layers][B][C] = 0 or 1/(1-p) per (sample, channel) from
 * Philox4x32-10 keyed on `seed`, counter (sample*C + channel, layers[i] | encoder << 16, step, rank) -- reproducible from those
 * numbers alone (the reference uses torch's global RNG stream). layers: HOST array of n_layers (<= 3) layer indices. */
int witw_dropout2d_scales(float* out, unsigned long long seed, unsigned encoder, unsigned step, unsigned rank, const int* layers,
                          int n_layers, int B, int C, float p, void* stream);

/* ---- triplet_loss, model/cvig_fov.py:366-382. workspace: 4*B floats, filled by fwd, read by bwd. */
int witw_triplet_loss_fwd(const float* distance /*[B,B]*/, int B, float alpha, float* loss /*[1]*/, float* workspace,
                          void* stream);
/* sharded global batch: un-normalised partial over the column slab D[Bo][Bs] (surfaces col0..col0+Bs of the global
 * batch), diag[Bo] = global diagonal; sum over ranks / (2*Bo*(Bo-1)) = the loss. workspace: Bo floats. */
int witw_triplet_loss_slab_fwd(const float* distance, const float* diag, int Bo, int Bs, int col0, float alpha, float* partial,
                               float* workspace, void* stream);
/* Backward of the sharded global-batch loss (the gradient of model/cvig_fov.py:366-382 restricted to one rank's column slab):
 * step 1 -> rowsig[Bo] = sum over THIS rank's surfaces of sigmoid(alpha*(d_ii - d_ij)) (the caller all-reduces it) and
 * colsig[Bs] = sum over all overheads of sigmoid(alpha*(d_cc - d_ij)), c = col0 + j (complete);
 * step 2 -> grad_distance[Bo,Bs] for the device scalar grad_loss of the global loss (normaliser 2*Bo*(Bo-1)). */
int witw_triplet_loss_slab_sig(const float* distance, const float* diag, int Bo, int Bs, int col0, float alpha, float* rowsig,
                               float* colsig, void* stream);
int witw_triplet_loss_slab_bwd(const float* distance, const float* diag, const float* rowsig, const float* colsig,
                               const float* grad_loss, float* grad_distance, int Bo, int Bs, int col0, float alpha, void* stream);
int witw_triplet_loss_bwd(const float* distance, const float* workspace, const float* grad_loss /*[1]*/,
                          float* grad_distance /*[B,B]*/, int B, float alpha, void* stream);

/* ---- bf16 inference path of the encoder (BASELINE config "cvig_semantic ... bf16 MFMA"): bf16 NHWC activations
 * (channels padded to 16) and packed bf16 filters, fp32 accumulate on v_mfma_f32_32x32x16_bf16, fp32 bias;
 * the last layer writes the fp32 NCHW embedding (out_nchw_f32). Pointers typed void* carry bf16 data. */
/* MFMA shape of the bf16 inference forward where both kernels apply (Cout >= 128, stride 1, >= 512 workgroups of 8 waves,
 * Cin % 32 == 0, bf16 NHWC out, no gate / Dropout2d scale / pool codes): 1 = v_mfma_f32_16x16x32_bf16 (default; two taps of a
 * 16-channel chunk per MFMA, the ninth tap of an even chunk shares its MFMA with the ninth tap of the next chunk), 0 =
 * v_mfma_f32_32x32x16_bf16. enable < 0 only queries; returns the previous setting. Results agree to one bf16 ulp. */
int witw_conv3x3_bf16_mfma16(int enable);
/* 1 (default): plain bf16 forwards with 64 input channels (layer 5 of the trunk, model/cvig_fov.py:261-262) and H % 8 == 0,
 * W % 32 == 0, Cout % 64 == 0, >= 4096 (8 x 32 tile, 64-channel block) units run on the kernel that keeps the filter block in
 * LDS (csrc/conv3x3_bf16_wres.hip); 0: on the tiled kernels. enable < 0 only queries; returns the previous setting. Bit-identical
 * to the 32x32x16 tiled kernel. */
int witw_conv3x3_bf16_wres(int enable);
long long witw_conv3x3_bf16_packed_elems(int cout, int cin);
int witw_conv3x3_bf16_pack_weights(const float* w_kcrs, void* wpk_bf16, int cout, int cin, void* stream);
/* transpose_flip != 0: the dgrad filter of the source tensor [cin][cout][3][3] (cout, cin describe the packed filter) */
int witw_conv3x3_bf16_pack_weights_ex(const float* w_kcrs, void* wpk_bf16, int cout, int cin, int transpose_flip, void* stream);
/* n filter images (and their biases) in one launch per 16 entries: what a bf16 training step re-packs after every Adam update
 * (csrc/pack_multi.hip). Entry i as witw_conv3x3_bf16_pack_weights_ex(w[i], wpk[i], cout[i], cin[i], transpose[i]); bias[i] NULL
 * or fp32 [cout[i]], copied to the first cout[i] floats of bias_dst[i]. Same bits as the one-at-a-time entry. */
int witw_conv3x3_bf16_pack_weights_multi(const void* const* w, void* const* wpk, const void* const* bias, void* const* bias_dst,
                                         const int* cout, const int* cin, const int* transpose, int n, void* stream);
int witw_nchw_f32_to_nhwc_bf16(const float* x, void* y_bf16, int B, int C, int H, int W, int Cpad, void* stream);
int witw_conv3x3_bf16_fwd(const void* x_bf16, const void* wpk_bf16, const float* bias, void* y, int B, int H, int W, int Cin,
                          int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, void* stream);

/* ---- fp32-grade inference on the fp16 MFMA ("fp16x3"): every value is carried as hi = fp16(x), lo = fp16(x - hi) and a
 * product is hi*hi + lo*hi + hi*lo with fp32 accumulation (3 MFMAs of the 2.5 PF/s pipe per fp32-equivalent product; the
 * encoder stays within 2e-5 of fp64, like the fp32 MFMA path, inside the reference goldens' 1e-4). Activations are NHWC
 * "split-fp16": per pixel and 8 channels 16 B of hi then 16 B of lo ([B,H,W,C/8,2,8] fp16, 4 B per value). Same layer
 * arguments as witw_conv3x3_bf16_fwd; the last layer writes the fp32 NCHW embedding (out_nchw_f32). */
long long witw_conv3x3_f16x3_packed_elems(int cout, int cin);
int witw_conv3x3_f16x3_pack_weights(const float* w_kcrs, void* wpk_f16, int cout, int cin, void* stream);
int witw_nchw_f32_to_split_f16(const float* x, void* y_split, int B, int C, int H, int W, int Cpad, void* stream);
int witw_split_f16_to_f32(const void* x_split, float* y /*[pixels][C]*/, long long pixels, int C, void* stream);
int witw_conv3x3_f16x3_fwd(const void* x_split, const void* wpk_f16, const float* bias, void* y, int B, int H, int W, int Cin,
                           int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, void* stream);
/* training forms, as witw_conv3x3_bf16_fwd_ex / witw_conv3x3_bf16_pack_weights_ex: Dropout2d scale, ReLU gate (a split-fp16
 * tensor shaped like y), zero-interleaved rows; transpose_flip packs the dgrad filter of the source [cin][cout][3][3] */
int witw_conv3x3_f16x3_pack_weights_ex(const float* w_kcrs, void* wpk_f16, int cout, int cin, int transpose_flip, void* stream);
int witw_conv3x3_f16x3_fwd_ex(const void* x_split, const void* wpk_f16, const float* bias, const float* dropmask,
                              const void* gate_split, void* y, unsigned char* pool_code, int* overflow_flag, int B, int H, int W,
                              int Cin, int Cout, int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, int dilate_h,
                              void* stream);
/* pool_code (uint8 [B,Hy,Wy,Cout]) as in witw_conv3x3_bf16_fwd_ex; overflow_flag: NULL or a device int set to 1 when a
 * split-fp16 output leaves the fp16 range (|v| > 65504 or NaN cannot be carried as hi + lo). The max-pool backward on split tensors: */
int witw_maxpool2x2_bwd_split(const void* dy_split, const unsigned char* code, void* dx_split, int B, int Hp, int Wp, int H, int W,
                              int C, void* stream);

/* weight gradient of the fp16x3 training step (csrc/wgrad_f16x3.hip): operands in the batch-octet split layout
 * [ceil(B/8)][H][W][C][2][8] fp16 (witw_split_f16_to_octet from split-fp16 NHWC), products hi*hi + lo*hi + hi*lo on the fp16
 * MFMA; dz_split = the gradient as split-fp16 NHWC (bias gradient; NULL with db NULL); dw / db fp32. */
long long witw_octet_split_elems(int B, int H, int W, int C);
int witw_split_f16_to_octet(const void* x_split, void* y_oct, int B, int H, int W, int C, void* stream);
long long witw_conv3x3_wgrad_f16x3_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h);
int witw_conv3x3_wgrad_f16x3(const void* x_oct, const void* dz_oct, const void* dz_split, float* dw, float* db, float* workspace,
                             int B, int H, int W, int Cin, int cin_real, int Cout, int stride_h, int pad_circular, int accumulate,
                             void* stream);

/* ---- bf16 (mixed-precision) TRAINING step of the encoder: the backward of the reference's training loop
 * (model/cvig_fov.py:447-460, autograd through torch.nn.Conv2d) with bf16 MFMA operands, fp32 accumulate, fp32
 * weight gradients / master weights / Adam.
 * witw_conv3x3_bf16_fwd_ex = witw_conv3x3_bf16_fwd plus: dropmask [B,Cout] fp32 (Dropout2d scale before the ReLU,
 * :287-288) or NULL; gate_bf16 = bf16 tensor shaped like y or NULL (outputs where gate <= 0 are zeroed: the ReLU
 * backward of a dgrad launch, which is this kernel on witw_conv3x3_bf16_pack_weights of the transposed, tap-rotated
 * filter); dilate_h = x holds (H-1)/2+1 physical rows standing for H zero-interleaved rows (dgrad of a stride-(2,1) layer);
 * pool_code = NULL or uint8 [B,Hy,Wy,Cout]: arg-max position (dy*2+dx) of the fused max-pool, consumed by
 * witw_maxpool2x2_bwd_bf16 (dy bf16 [B,Hp,Wp,C] -> dx bf16 [B,H,W,C]). */
int witw_conv3x3_bf16_fwd_ex(const void* x_bf16, const void* wpk_bf16, const float* bias, const float* dropmask,
                             const void* gate_bf16, void* y, unsigned char* pool_code, int B, int H, int W, int Cin, int Cout,
                             int stride_h, int pad_circular, int relu, int pool, int out_nchw_f32, int dilate_h, void* stream);
int witw_maxpool2x2_bwd_bf16(const void* dy_bf16, const unsigned char* code, void* dx_bf16, int B, int Hp, int Wp, int H, int W,
                             int C, void* stream);
/* Weight gradient on v_mfma_f32_32x32x16_bf16. Both operands in the batch-octet layout [ceil(B/8)][H][W][C][8 images]
 * (witw_nhwc_bf16_to_octet; witw_octet_elems = its element count): the 8 k values of an MFMA lane are 8 images at one
 * pixel, so a filter tap is a pixel offset and no transposition is needed. x_oct = the layer's input, dz_oct = the
 * gradient at its output; dw [Cout][cin_real][3][3] fp32, db [Cout] fp32 or NULL (summed inside the same kernel);
 * workspace of witw_conv3x3_wgrad_bf16_workspace_floats floats. */
long long witw_octet_elems(int B, int H, int W, int C);
int witw_nhwc_bf16_to_octet(const void* x_bf16, void* y_bf16, int B, int H, int W, int C, void* stream);
long long witw_conv3x3_wgrad_bf16_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h);
int witw_conv3x3_wgrad_bf16(const void* x_oct, const void* dz_oct, float* dw, float* db, float* workspace, int B, int H, int W,
                            int Cin, int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, void* stream);
/* Round 5: the same gradient straight from the NHWC bf16 tensors the forward / dgrad launches write (x [B][H][W][Cin], dz
 * [B][Ho][W][Cout]) -- no re-layout pass: the MFMA's k index is the pixel, the [pixel][channel] LDS image is transposed on the
 * way out by ds_read_b64_tr_b16 (replaces autograd through torch.nn.Conv2d in model/cvig_fov.py:447-460 like the entry above,
 * and is what the bf16 training step now calls). Workspace of witw_conv3x3_wgrad_bf16_nhwc_workspace_floats floats. */
long long witw_conv3x3_wgrad_bf16_nhwc_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride_h);
/* 1: eligible layers (stride 1, Cin > 32, Cout > 64, W % 32 == 0) run it on v_mfma_f32_16x16x32_bf16; 0 (default; that form
 * measured 3-9 % slower): v_mfma_f32_32x32x16_bf16. enable < 0 only queries; returns the previous setting. Set it BEFORE sizing the
 * workspace. Same parity bar for both. */
int witw_conv3x3_wgrad_bf16_mfma16(int enable);
int witw_conv3x3_wgrad_bf16_nhwc(const void* x_nhwc, const void* dz_nhwc, float* dw, float* db, float* workspace, int B, int H, int W,
                                 int Cin, int cin_real, int Cout, int stride_h, int pad_circular, int accumulate, void* stream);

/* ---- 2x2 sub-window form of the 3x3 kernels: cvig_baseline's Conv2d(k=4,s=2,p=0) (model/cvig_baseline.py:236-252) is a
 * 2x2 convolution over the space-to-depth(2) image, i.e. a 3x3 window whose first tap row/column are zero (dgrad: last
 * row/column); these entries run only the 4 live taps (2.25x fewer MFMAs than witw_conv3x3_fwd_ex on the zero-filled
 * 3x3 filter, same results). pack: taps {1,2}^2 of w [cout][cin][3][3] (transpose_flip: taps {0,1}^2 of the dgrad
 * filter of the source [cin][cout][3][3]); fwd: tap_base = 1 with a forward-packed filter, 0 with a dgrad-packed one;
 * wgrad: taps {1,2}^2 of dw, exact zeros elsewhere (workspace as witw_conv3x3_wgrad, stride 1). */
long long witw_conv3x3_packed_floats_taps4(int cout, int cin);
int witw_conv3x3_pack_weights_taps4(const float* w_kcrs, float* wpk, int cout, int cin, int transpose_flip, void* stream);
int witw_conv3x3_fwd_taps4(const float* x, const float* wpk4, const float* bias, const float* gate, const float* post_scale,
                           const float* post_shift, float* y, int B, int H, int W, int Cin, int Cout, int relu,
                           float lrelu_slope, int tap_base, void* stream);
/* The same convolution with two options for the deep, small-map layers of the encoder (model/cvig_baseline.py:240-252, conv5-7:
 * 16x16 .. 4x4 maps with K = 4 taps x 2048 channels start far fewer workgroups than the chip has CUs):
 *  ksplit > 1  split-K: y is a workspace of ksplit*B*H*W*Cout floats receiving the RAW partial sums of each K slice
 *              ([ksplit][B,H,W,Cout]; no bias / activation / affine; gate must be NULL); witw_taps4_splitk_finish reduces them.
 *              witw_conv3x3_taps4_ksplit returns the factor the library would pick (1 = the plain launch fills the chip).
 *  s2d != 0    (ksplit == 1) y is written as the space-to-depth(2) image of the valid_h x valid_w region,
 *              [B, ceil(valid_h/2), ceil(valid_w/2), 4*Cout], zeros outside it: the layout the next k=4,s=2 layer reads
 *              (replaces a witw_space_to_depth2 pass over the activation). */
int witw_conv3x3_taps4_ksplit(int B, int H, int W, int Cin, int Cout);
int witw_conv3x3_fwd_taps4_ex(const float* x, const float* wpk4, const float* bias, const float* gate, const float* post_scale,
                              const float* post_shift, float* y, int B, int H, int W, int Cin, int Cout, int relu,
                              float lrelu_slope, int tap_base, int ksplit, int s2d, int valid_h, int valid_w, void* stream);
int witw_conv3x3_wgrad_taps4(const float* x, const float* dz, float* dw, float* db, float* workspace, int B, int H, int W,
                             int Cin, int cin_real, int Cout, int accumulate, void* stream);

/* ---- cvig_baseline (model/cvig_baseline.py). Conv2d(k=4,s=2,p=0) = the 3x3 kernel above on the
 * space-to-depth(2) image with a filter whose first tap row/column is zero. */
/* x NHWC [B,Hp,Wp,C] (NCHW if in_nchw) with valid region HxW -> NHWC [B,ceil(H/2),ceil(W/2),Cpad], channel
 * (dy*2+dx)*C+c; normalize=1 applies x/255, -1+2x (:265-266); scale/shift (NULL or [C]): per-channel affine of a
 * train-mode BatchNorm applied on the fly (also accepted by witw_gem_pool). */
/* The FIRST block of an encoder in one launch (model/cvig_baseline.py:236-240 conv1/bn1, :265-268 the in-model x/255, -1+2x,
 * LeakyReLU, eval-mode BatchNorm): x NCHW fp32 [B,C<=5,H,W] -> y [B, ceil(vh/2), ceil(vw/2), 256], the space-to-depth(2) image of
 * the vh x vw = ((H-4)/2+1) x ((W-4)/2+1) valid outputs (zeros elsewhere) that the second block reads. w: torch layout
 * [64][C][4][4]; scale / shift [64] (both or neither). Replaces witw_space_to_depth2(in_nchw, normalize) +
 * witw_conv3x3_fwd_taps4_ex(s2d = 1) for that block. */
int witw_conv4x4s2_first_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y,
                             int B, int C, int H, int W, int normalize, float lrelu_slope, void* stream);
int witw_space_to_depth2(const float* x, float* y, int B, int Hp, int Wp, int H, int W, int C, int Cpad, int in_nchw,
                         int normalize, const float* scale, const float* shift, void* stream);
/* Mosaic space-to-depth: x NHWC [B,H,W,C] (C % 4 == 0) -> y [ceil(B/g^2), g*ceil(H/2), g*ceil(W/2), 4C]: image b is cell b % g^2
 * (row-major) of mosaic image b / g^2. A k=4,s=2 layer over an n x n s2d map has (n-1)^2 valid outputs, so no valid window crosses
 * a cell: conv6 / conv7 (8x8, 4x4 maps) run as g = 2 / 4 mosaics that fill the kernel's 16x16 tile. */
int witw_space_to_depth2_mosaic(const float* x, float* y, int B, int H, int W, int C, int g, void* stream);
/* Reduce the split-K partial sums ws [ksplit][ceil(B/g^2), g*h, g*w, C] in slice order and apply bias, activation (0 none, 1 ReLU,
 * 2 LeakyReLU) and the optional per-channel affine: y [B,vh,vw,C] = the valid outputs of every image. */
int witw_taps4_splitk_finish(const float* ws, int ksplit, const float* bias, int act, float lrelu_slope, const float* post_scale,
                             const float* post_shift, float* y, int B, int g, int h, int w, int vh, int vw, int C, void* stream);
/* f[b,col0+c] = (mean relu(x)^p)^(1/p) over the valid HxW region (:276-282); f has row length ldf. */
int witw_gem_pool(const float* x, float* f, int B, int Hp, int Wp, int H, int W, int C, int ldf, int col0, float p,
                  const float* scale, const float* shift, void* stream);
/* f[b,:] /= sqrt(|f[b,:]|_2) in place (:284) */
int witw_embed_normalize(float* f, int B, int n, void* stream);
/* D[i][j] = |a_i - b_j|^2 (or its square root: Euclidean ranking distance of :457-458) */
int witw_pairwise_sqdist(const float* a, const float* b, float* D, int Na, int Nb, int n, int take_sqrt, void* stream);
/* exhaustive_minibatch_triplet_loss (:286-315) from D[i][j] = |embed1_i - embed2_j|^2; workspace B floats */
int witw_exhaustive_triplet_loss(const float* D, int B, int soft_margin, float alpha, float margin, float* loss,
                                 float* workspace, void* stream);

/* training mode of cvig_baseline (BatchNorm2d batch statistics, autograd of :267-315) */
long long witw_bn_workspace_floats(int B, int H, int W, int C);
/* batch mean / biased variance over the valid HxW region of a [B,Hp,Wp,C] tensor -> mean, invstd, and the affine
 * y = a*scale + shift; running_mean/var (may be NULL) are updated with `momentum` and the unbiased variance. */
int witw_bn_train_stats(const float* a, int B, int Hp, int Wp, int H, int W, int C, const float* gamma, const float* beta,
                        float eps, float momentum, float* mean, float* invstd, float* scale, float* shift, float* running_mean,
                        float* running_var, float* workspace, void* stream);
/* backward of y = BatchNorm_train(LeakyReLU(z)): a = LeakyReLU(z) as saved, dy at y -> dz (0 outside the valid region),
 * dgamma, dbeta. */
int witw_bn_lrelu_bwd(const float* a, const float* dy, float* dz, float* dgamma, float* dbeta, const float* mean,
                      const float* invstd, const float* gamma, int B, int Hp, int Wp, int H, int W, int C, float slope,
                      float* workspace, void* stream);
/* the same with dy still in the space-to-depth layout the next block's data-gradient conv wrote ([B, ceil(H/2), ceil(W/2), dy_s2d_cp],
 * channel ((h&1)*2+(w&1))*C + c), read in place: saves the witw_depth_to_space2 pass; dy_s2d_cp = 0: dy is [B,Hp,Wp,C] */
int witw_bn_lrelu_bwd_ex(const float* a, const float* dy, float* dz, float* dgamma, float* dbeta, const float* mean,
                         const float* invstd, const float* gamma, int B, int Hp, int Wp, int H, int W, int C, float slope,
                         int dy_s2d_cp, float* workspace, void* stream);
/* inverse of witw_space_to_depth2 for gradients: g [B,ceil(H/2),ceil(W/2),Cpad] -> dx [B,Hp,Wp,C] (+ add, may be NULL) */
int witw_depth_to_space2(const float* g, const float* add, float* dx, int B, int Hp, int Wp, int H, int W, int C, int Cpad,
                         void* stream);
int witw_gem_pool_bwd(const float* a, const float* scale, const float* shift, const float* f, const float* df, float* dy, int B,
                      int Hp, int Wp, int H, int W, int C, int ldf, int col0, float p, int accumulate, void* stream);
/* g: un-normalised feature [B,n], df: gradient at g/sqrt(|g|) -> dg */
int witw_embed_normalize_bwd(const float* g, const float* df, float* dg, int B, int n, void* stream);
/* backward of witw_pairwise_sqdist + witw_exhaustive_triplet_loss; workspace B*B floats */
int witw_exhaustive_triplet_loss_bwd(const float* e1, const float* e2, const float* D, const float* grad_loss, float* de1,
                                     float* de2, int B, int n, int soft_margin, float alpha, float margin, float* workspace,
                                     void* stream);

/* ---- data path: Resize (:100-134), ImageNormalization (:137-149; semantic: cvig_semantic.py:167-176),
 *      PolarTransform (:156-209). NCHW fp32. mean/stdv: HOST arrays of C floats (NULL = resize only). */
int witw_resize_bilinear_normalize(const float* x, float* y, int B, int C, int Hi, int Wi, int Ho, int Wo,
                                   const float* mean, const float* stdv, int n_div255, void* stream);
/* Resize (+ normalise) a batch of images of INDIVIDUAL sizes in one launch (the DataLoader hands over raw images of any size,
 * model/cvig_fov.py:393-403). desc: DEVICE array [B][5] of 64-bit words {device address of image b, H, W, start column, channels
 * stored per pixel}; kind 0 = float32 planar CHW sources, 1 = uint8 interleaved HWC (decoder output). y [B,C,Ho,Wo] = the image
 * resized to Ho x Wfull, columns start .. start+Wo-1 (mod Wfull) kept: the panorama crop of Resize (:118-128); Wfull == Wo and
 * start == 0 give the plain resize. Same arithmetic as witw_resize_bilinear_normalize. */
int witw_resize_bilinear_normalize_batched(const void* desc, float* y, int B, int C, int Ho, int Wo, int Wfull, int kind,
                                           const float* mean, const float* stdv, int n_div255, void* stream);
int witw_normalize(const float* x, float* y, int B, int C, int H, int W, const float* mean, const float* stdv, int n_div255,
                   void* stream);
/* taps: int32 [Ho*Wo][4] flat offsets into a size*size plane; wts: fp32 [Ho*Wo][4]; both DEVICE tables
 * built on the host in fp64 exactly as model/cvig_fov.py:163-181,197-201. */
int witw_polar_transform(const float* x, const int* taps, const float* wts, float* y, int B, int C, int size, int Ho, int Wo,
                         void* stream);
/* Overhead side of Compose[Resize, ImageNormalization, PolarTransform] (model/cvig_fov.py:393-397, classes :100-209) in ONE
 * launch: raw image -> bilinear size x size value -> (x/255 - mean)/std -> 4-tap polar gather -> y [B,C,Ho,Wo]; the size x size
 * image is never written; bit-identical to witw_resize_bilinear_normalize(_batched) followed by witw_polar_transform.
 * desc == NULL: src = fp32 [B,C,Hi,Wi]; else desc = the DEVICE table of witw_resize_bilinear_normalize_batched (kind 0 fp32
 * CHW, 1 uint8 HWC). wts: the weights of witw_polar_transform's table; taps: its taps as offsets (row * box width + column)
 * inside their tile's box. tiles: DEVICE int32 [n_tile][8] = {box x0, y0, width,
 * height, first output row, first output column, rows, columns}: a partition of the Ho x Wo outputs into tiles of at most 256
 * outputs, each with the bounding box (in the size x size plane) of all taps it reads, box width and height <= 64;
 * max_box = the largest box width*height (-1 otherwise: use the separate launches). */
int witw_polar_from_raw(const void* src, const void* desc, int kind, float* y, int B, int C, int Hi, int Wi, int size, int Ho,
                        int Wo, const int* taps, const float* wts, const int* tiles, int n_tile, int max_box, const float* mean,
                        const float* stdv, int n_div255, void* stream);
/* bilinear_interpolate (model/cvig_fov.py:156-183) on arbitrary coordinates: the same 4-tap gather with a caller-built
 * table; taps = flat offsets into a plane of plane_in elements, n_out samples per plane: y [B,C,n_out]. */
int witw_bilinear_gather(const float* x, const int* taps, const float* wts, float* y, int B, int C, long long plane_in,
                         long long n_out, void* stream);
/* SyncedRotation's overhead rotation (model/cvig_baseline.py:142: torchvision.transforms.functional.rotate on a float
 * CHW tensor = nearest-neighbour affine grid sample, zero fill, same size). x,y [B,C,H,W] fp32, y != x; theta DEVICE
 * fp32 [B][3][2]: per image the inverse rotation matrix divided by (W/2, H/2), row k = (x, y, 1) coefficient. */
int witw_rotate_nearest(const float* x, const float* theta, float* y, int B, int C, int H, int W, void* stream);

/* Conv2d(k=4, s=2) filter [co][ci][4][4] <-> the 3x3 filter over space-to-depth(2) channels [co][cpad][3][3] the taps4 kernels take
 * (model/cvig_baseline.py:236-252; tap row 0 / column 0 zero, channel (dy*2+dx)*ci + c); the second call is the inverse gather for
 * weight gradients. */
int witw_conv4x4_to_k3(const float* w, float* k3, int co, int ci, int cpad, void* stream);
int witw_k3_to_conv4x4(const float* k3, float* w, int co, int ci, int cpad, void* stream);

/* ---- JPEG back end on the device: what libjpeg does behind entropy decoding for the images the reference reads
 *      (skimage.io.imread -> PIL -> libjpeg-turbo, model/cvig_fov.py:88-89, in the DataLoader workers of :402). The entropy
 *      decoding stays on the host (witw_amd/csrc_host/jpeg_coef.cpp -> libwitw_jpeg.so: witw_jpeg_info, witw_jpeg_decode_coef).
 *      Byte-identical to Pillow's decode (JDCT_ISLOW, fancy upsampling).
 * witw_jpeg_idct: dequantisation + 'islow' integer inverse DCT. coef: DEVICE int16 [total_blocks][64] natural order; qt: DEVICE
 * uint16 [tables][64]; planes: DEVICE int64 [n_planes][6] = {first block, table index, byte offset of the output plane, blocks
 * wide, blocks high, blocks of all earlier planes}; out: the 8-bit component planes (bh*8 rows of bw*8 bytes). */
int witw_jpeg_idct(const void* coef, const void* qt, const void* planes, int n_planes, long long total_blocks, void* out, void* stream);
/* witw_jpeg_to_rgb: fancy chroma upsampling + YCbCr -> RGB. planes: the output above; images: DEVICE int64 [n_images][12] =
 * {H, W, components (1 | 3), mode (0 none, 1 h2v1, 2 h2v2 fancy, 3 / 4 the same replicated: chroma planes of <= 2 columns), luma offset, luma stride, Cb offset, Cr offset, chroma stride, chroma
 * rows, chroma columns (real samples), output offset}; out: H x W x components interleaved bytes per image. */
int witw_jpeg_to_rgb(const void* planes, const void* images, int n_images, long long max_pixels, void* out, void* stream);
/* Entropy decoding ON THE DEVICE for files that carry restart markers (the reference decodes in its loader workers,
 * model/cvig_fov.py:88-89, :402): one GPU thread per restart interval. files: DEVICE int64 [n_files][4] = {address of the file bytes
 * (8-byte aligned, 24 readable bytes behind the end), address of the file's plan (host side: witw_jpeg_entropy_plan in
 * libwitw_jpeg.so -- header fields, Huffman tables, byte offset of every interval; a byte scan, no Huffman decoding), address of the
 * file's coefficient area int16 [blocks][64] (zero-filled by the caller), file length}; max_intervals: the largest interval count of a
 * file of the launch (grid sizing: one workgroup per 256 intervals of a file); errors: DEVICE int32 [n_files], zeroed by the
 * caller, 1 where a file's entropy-coded data is damaged, 2 for a bad plan. Coefficients bit-identical to the host decoder's
 * (witw_jpeg_decode_coef); witw_jpeg_idct / witw_jpeg_to_rgb take it from there. */
int witw_jpeg_huffman(const void* files, int n_files, int max_intervals, int* errors, void* stream);
/* The same for files WITHOUT restart markers (their plan holds ONE interval: the whole scan): a self-synchronising decode, one
 * workgroup of 512 threads per file (WITW_SELFSYNC_THREADS = 256 | 512 | 1024) -- the scan is unstuffed and cut into as many subsequences, every thread decodes its own from a
 * guessed state, then from its neighbour's exit state, round after round until no entry state changes (Huffman codes fall into step
 * after a few dozen symbols; thread 0 starts at the true state, so the fixed point is the true decoding); a prefix sum numbers the
 * blocks, a last pass writes the coefficients and a per-component prefix sum turns DC differences into DC values. files: DEVICE int64
 * [n_files][6] = {file bytes, plan, coefficient area (zero-filled), file length, scratch of file length + 32 bytes (8-byte aligned),
 * 0}. Coefficients bit-identical to the host decoder's. */
int witw_jpeg_huffman_selfsync(const void* files, int n_files, int* errors, void* stream);
/* The same with the threads per file chosen by the caller (256, 512 or 1024; jpeg.decode_packed_multi: 1024 when the launch holds a file
 * of 48 KB or more, else 512 -- larger files gain from shorter subsequences, smaller ones lose to the extra rounds). */
int witw_jpeg_huffman_selfsync_threads(const void* files, int n_files, int threads, int* errors, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WITW_HIP_H */
