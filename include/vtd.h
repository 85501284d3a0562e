/*
 * vtd.h -- C ABI of libvtd_hip.so, the MI355X (gfx950) implementation of the per-frame
 * text-detection / recognition hot path of malak29/video-text-detection-system.
 *
 * The reference has no FFI for this path: its boundary is the Python class API of app/ml
 * (TextDetector / TextRecognizer / VideoTextPipeline).  The entry points below are what a binding
 * for those classes needs; each cites the reference interface it replaces (file:line relative to the
 * reference repository).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain C, opaque handles, caller-owned device buffers, no torch / C++ types in signatures
 *   - every call returns 0 on success, a negative value on failure: -(hipError_t) for HIP errors,
 *     -1000 and below for argument / shape validation errors; vtd_strerror() names them
 *   - every launching call takes the HIP stream to enqueue on (pass NULL for the default stream);
 *     nothing synchronises the device unless documented
 *   - no process-global state: N handles (one per GPU / per worker thread) coexist; a handle must be
 *     used from one stream at a time
 *   - "dev" pointers are device memory on the handle's device; "host" pointers are host memory
 */
#ifndef VTD_H
#define VTD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vtd_detector vtd_detector;
typedef struct vtd_recognizer vtd_recognizer;
typedef struct vtd_postproc vtd_postproc;
typedef void* vtd_stream; /* hipStream_t */

/* One detection, as TextDetector._post_process emits it (app/ml/models/text_detector.py:172-176):
 * bbox in frame pixels, polygon (4 x (x,y)) in map space, confidence = mean probability. */
typedef struct vtd_detection {
    int32_t bbox[4];
    int32_t polygon[8];
    float confidence;
    float area;            /* contour area of the component (diagnostic) */
    int32_t first_x, first_y; /* raster-first pixel of the component (diagnostic / ordering) */
} vtd_detection;

/* ---- library ---------------------------------------------------------------------------------- */
const char* vtd_version(void);
const char* vtd_strerror(int code);
/* number of visible HIP devices, or a negative error */
int vtd_device_count(void);
/* A HIP stream whose kernels are confined to the CUs named by cu_mask (bit i of word i / 32 = CU i; hipExtStreamCreateWithCUMask).
 * The Transformer recogniser runs its encoder pass and its greedy decode on two such streams with disjoint masks (vtd_amd/engine.py:
 * TrOCREngine), so the next batch's encoder pass overlaps the current batch's decode.  Destroy with vtd_stream_destroy. */
int vtd_stream_create_masked(const uint32_t* cu_mask, int words, vtd_stream* out);
int vtd_stream_destroy(vtd_stream stream);

/* ---- detector: DBNet (text_detector.py:12-86) -------------------------------------------------- */
/* backbone: "resnet18" | "resnet50" (text_detector.py:16-20; 'resnet18' is the documented repair A2).
 * max_batch frames of 640x640 network input are provisioned in HBM at creation. */
int vtd_detector_create(const char* backbone, int max_batch, vtd_detector** out);
void vtd_detector_destroy(vtd_detector* d);
/* Feed one tensor of the reference checkpoint's model_state_dict (text_detector.py:108-109) by its key,
 * e.g. "backbone.0.weight", "fpn.inner_blocks.2.bias", "head.probability_head.3.weight"; float32,
 * PyTorch memory order.  Unknown keys are rejected, "num_batches_tracked" keys are accepted and ignored. */
int vtd_detector_set_tensor(vtd_detector* d, const char* key, const float* host_data, int64_t numel);
/* Build options, before finalize.  "fuse_fpn_head" (default 1): evaluate FPN lateral(C2) + top-down add + P2 smooth +
 * head conv as one algebraically composed convolution (the 256-channel P2 map is never formed; its "p2" tap is then
 * unavailable).  0 keeps the layer-by-layer graph.
 * "fuse_stem_pool" (default 1): backbone conv1 7x7/s2 + bn1 + relu + maxpool 3x3/s2 run as one kernel that only writes the
 * pooled map (tap "pool"); 0 runs the generic convolution + a separate pool kernel and exposes the tap "stem".
 * "fuse_downsample" (default 1): the 1x1 projection + BatchNorm on the residual branch of a ResNet downsample block
 * (text_detector.py:16-19 -> torchvision BasicBlock / Bottleneck `downsample`) is evaluated inside the block's last
 * convolution as extra K-steps over the block input (the projected map is never written); 0 runs it as its own launch. */
int vtd_detector_set_option(vtd_detector* d, const char* name, int value);
/* Folds BatchNorm, repacks to the kernels' fp16 layouts and uploads.  Fails (-1103) if a key is missing. */
int vtd_detector_finalize(vtd_detector* d, vtd_stream stream);
/* K1 = cvtColor(BGR2RGB) + ToPILImage + Resize((640,640)) + ToTensor + Normalize (text_detector.py:99-104,
 * 119-124) for n frames of identical size: frames_dev is [n,H,W,3] uint8 BGR.  Fills the network input. */
int vtd_detector_preprocess(vtd_detector* d, const uint8_t* frames_dev, int n, int height, int width, vtd_stream stream);
/* Alternative input: the tensor the reference hands to DBNet.forward, [n,3,640,640] float32 (text_detector.py:124). */
int vtd_detector_set_input_nchw(vtd_detector* d, const float* x_dev, int n, vtd_stream stream);
/* DBNet.forward (text_detector.py:25-29) on the current input: prob_dev receives [n,640,640] float32
 * probabilities (the 'probability' map); thresh_dev, if not NULL, the 'threshold' map. */
int vtd_detector_forward(vtd_detector* d, int n, float* prob_dev, float* thresh_dev, vtd_stream stream);
/* Debug / test taps: copies an internal activation as dense NCHW float32 to host.  name: "input", "c2".."c5",
 * "p2", "stem", "head1", "head2".  Synchronises the stream. */
int vtd_detector_read_tap(vtd_detector* d, const char* name, int n, float* host_out, int64_t capacity, vtd_stream stream);
/* Algorithmic live work of one frame through this detector, in MACs (for roofline accounting). */
int64_t vtd_detector_macs_per_frame(const vtd_detector* d);
/* Per-launch HIP-event timing on the launch stream (bench.py's roofline leg).  set_profiling(1) resets the
 * accumulators; while enabled every vtd_detector_forward brackets each of its launches with events (enable = 2 + k:
 * only launch slot k, two events per forward, so the timed region is not perturbed).
 * get_profile resolves pending events (synchronises the stream) and returns, for launch slot op_index in
 * [0, vtd_detector_num_ops), a description, the accumulated milliseconds, launch count and algorithmic MACs. */
int vtd_detector_set_profiling(vtd_detector* d, int enable);
int vtd_detector_num_ops(const vtd_detector* d);
int vtd_detector_get_profile(vtd_detector* d, int op_index, char* name, int name_cap, double* total_ms, int64_t* calls,
                             double* total_macs, vtd_stream stream);

/* Kernel-selection table.  Which tile configuration / kernel variant runs a convolution is decided per launch slot and
 * power-of-two batch bucket: from the table when it holds a valid entry, else by a timing contest on first use (whose result
 * joins the table).  set_tuning merges a table (text lines "<signature>|n<bucket> <config id>", '#' comments) -- the one shipped
 * with the package, or rank 0's, so that every process and every rank selects the same kernels and produces bit-identical
 * maps; get_tuning serialises the current table into buf (NUL-terminated, truncated to capacity) and returns the size
 * needed; tuning_measured tells whether any entry came from this process's own contest.  The reference has no counterpart
 * (ATen picks its kernels internally). */
int vtd_detector_set_tuning(vtd_detector* d, const char* table_text);
int64_t vtd_detector_get_tuning(const vtd_detector* d, char* buf, int64_t capacity);
int vtd_detector_tuning_measured(const vtd_detector* d);

/* ---- post-process: TextDetector._post_process (text_detector.py:143-178) ------------------------ */
/* Workspace for maps of map_h x map_w (the reference hard-codes 640 in the bbox arithmetic but its tests feed
 * 160x160 maps: any 2-D size works) and up to max_batch maps per call; max_out detections kept per frame. */
int vtd_postproc_create(int max_batch, int map_h, int map_w, int max_out, vtd_postproc** out);
void vtd_postproc_destroy(vtd_postproc* pp);
/* prob_dev: [n,map_h,map_w] float32.  orig_w/orig_h: frame sizes (host arrays of n).  Strict `p > threshold`.
 * out_dev: [n,max_out] records in the order cv2.findContours(RETR_EXTERNAL) yields contours (reverse raster
 * discovery); counts_dev[i] = number of detections of frame i before truncation to max_out. */
int vtd_postproc_run(vtd_postproc* pp, const float* prob_dev, int n, const int32_t* orig_w_host, const int32_t* orig_h_host,
                     float threshold, vtd_detection* out_dev, int32_t* counts_dev, vtd_stream stream);
/* Copy `bytes` of device memory to PINNED host memory (hipHostMalloc / torch pin_memory; both pointers 16-byte aligned) with a kernel on
 * `stream` instead of hipMemcpyAsync -- a launch never makes the host wait for the GPU, which the asynchronous copy of the detection records
 * was seen to do once per drained pipeline.  Order the host behind it with an event on `stream`.  Fails (HIP error) for pageable memory. */
int vtd_copy_to_pinned_host(const void* src_dev, void* dst_pinned_host, int64_t bytes, vtd_stream stream);

/* ---- training loss, forward only (app/ml/training/trainer.py:48-56 training_step, :66-71 validation_step, :130-142 DiceLoss) --
 * total = nn.BCELoss()(probability, probability_map) + nn.BCELoss()(threshold, threshold_map) + DiceLoss()(probability,
 * probability_map) over `numel` float32 elements per map (any shape; 16-byte aligned device pointers), in ONE pass that reads every
 * element once.  out4_dev = {probability BCE, threshold BCE, dice loss, total} as float32 (the reference's scalar tensors);
 * sums5_dev (optional) = the five float64 sums behind them (BCE numerators, sum p*t, sum p, sum t).  thresh_dev / thresh_target_dev
 * may both be null (DiceLoss alone, or a detector run without its threshold branch): that term is then 0.  smooth = DiceLoss.smooth
 * (1e-5).  workspace_dev: vtd_dbloss_workspace_bytes() bytes, caller-owned, one per concurrent call.  fp64 accumulation in a fixed
 * order: bitwise repeatable.  Backward, AdamW and the plateau scheduler (trainer.py:107-128) are not part of this library yet. */
int64_t vtd_dbloss_workspace_bytes(void);
int vtd_dbloss_forward(const float* prob_dev, const float* thresh_dev, const float* prob_target_dev, const float* thresh_target_dev, int64_t numel,
                       float smooth, void* workspace_dev, float* out4_dev, double* sums5_dev, vtd_stream stream);

/* ---- recogniser: CRNN (app/ml/models/text_recognizer.py:12-37,114-167) --------------------------- */
/* vocab_size = len(TextRecognizer.vocab) = 97 (text_recognizer.py:86-91); max_crops text regions per call. */
int vtd_recognizer_create(int vocab_size, int max_crops, vtd_recognizer** out);
void vtd_recognizer_destroy(vtd_recognizer* r);
/* One tensor of the CRNN checkpoint (text_recognizer.py:95-96): "cnn.N.*", "rnn.weight_ih_l0[_reverse]", ...,
 * "classifier.weight|bias"; float32, PyTorch memory order. */
int vtd_recognizer_set_tensor(vtd_recognizer* r, const char* key, const float* host_data, int64_t numel);
/* Build options, before finalize.  "fuse_pools" (default 1): the MaxPool2d((2,2)) / ((2,1)) layers behind conv2, conv4 and
 * conv6 + ReLU (text_recognizer.py:17-22) are taken in those convolutions' epilogues (the un-pooled maps are never written;
 * bit-identical results); 0 runs them as separate pool launches. */
int vtd_recognizer_set_option(vtd_recognizer* r, const char* name, int value);
int vtd_recognizer_finalize(vtd_recognizer* r, vtd_stream stream);
/* K6: for every box (frame, x1, y1, x2, y2) take frame[y1:y2, x1:x2] (pipeliine.py:121) out of frames_dev
 * ([n_frames,H,W,3] uint8 BGR) and cv2.resize it to 128x32 (text_recognizer.py:118), then run conv1.  boxes_dev is
 * [ncrops][5] int32 in device memory.  Boxes must lie inside the frame (invalid ones yield a zero crop). */
int vtd_recognizer_crop_resize(vtd_recognizer* r, const uint8_t* frames_dev, int n_frames, int height, int width,
                               const int32_t* boxes_dev, int ncrops, vtd_stream stream);
/* Alternative input: the tensor the reference hands to CRNN.forward, [ncrops,3,32,128] float32 (text_recognizer.py:122). */
int vtd_recognizer_set_input_nchw(vtd_recognizer* r, const float* x_dev, int ncrops, vtd_stream stream);
/* CRNN.forward (text_recognizer.py:29-37) on the current input: logits_dev receives [ncrops,31,vocab] float32. */
int vtd_recognizer_forward(vtd_recognizer* r, int ncrops, float* logits_dev, vtd_stream stream);
/* Test taps as dense NCHW float32 on the host: "resized" ([n,32,128,3] uint8 values), "cnn" ([n,512,1,31]),
 * "h0", "h1" ([n,512,1,31] LSTM layer outputs).  Synchronises the stream. */
int vtd_recognizer_read_tap(vtd_recognizer* r, const char* name, int ncrops, float* host_out, int64_t capacity, vtd_stream stream);
int64_t vtd_recognizer_macs_per_crop(const vtd_recognizer* r);
/* Kernel-selection table of the recogniser (see vtd_detector_set_tuning); buckets are powers of two of the crop count. */
int vtd_recognizer_set_tuning(vtd_recognizer* r, const char* table_text);
int64_t vtd_recognizer_get_tuning(const vtd_recognizer* r, char* buf, int64_t capacity);
int vtd_recognizer_tuning_measured(const vtd_recognizer* r);
/* [softmax(dim=2) +] TextRecognizer._decode_prediction (text_recognizer.py:126,142-167) for n sequences of
 * dense [T,V] rows (T <= 128): apply_softmax=1 takes logits and fuses the softmax, 0 takes probabilities as
 * _decode_prediction itself does.  id2char_dev[V]: code point per class id or -1 for ids that emit nothing
 * ('<blank>', '<unk>').  out_dev: [n][2+T] int32 = length, float confidence bits, then the code points. */
int vtd_ctc_greedy_decode(const float* logits_dev, int n, int T, int V, const int32_t* id2char_dev, int blank_id, int apply_softmax,
                          int32_t* out_dev, vtd_stream stream);

/* ---- Transformer recogniser: TrOCR (app/ml/models/text_recognizer.py:39-69) ----------------------------------------------
 * TransformerRecognizer loads VisionEncoderDecoderModel "microsoft/trocr-base-printed" (ViT encoder + TrOCR decoder) and calls
 * generate(pixel_values, max_length=50).  The architecture is a parameter (the checkpoint's config.json values; defaults of the
 * Python binding restate trocr-base-printed), weights arrive by their transformers-4.36 state-dict keys. */
typedef struct vtd_trocr vtd_trocr;
typedef struct vtd_trocr_config {
    int32_t image_size, patch_size;                       /* 384, 16 */
    int32_t enc_hidden, enc_layers, enc_heads, enc_ffn;   /* 768, 12, 12, 3072; head width must be 64 */
    int32_t enc_qkv_bias;                                 /* 0 for the TrOCR checkpoints */
    float enc_ln_eps;                                     /* 1e-12 */
    int32_t dec_hidden, dec_layers, dec_heads, dec_ffn;   /* 1024, 12, 16, 4096 */
    int32_t vocab_size, max_positions;                    /* 50265, 512 */
    float dec_ln_eps;                                     /* 1e-5 */
    int32_t decoder_start_token_id, eos_token_id, pad_token_id; /* 2, 2, 1 */
    int32_t max_length;                                   /* 50: text_recognizer.py:58 */
} vtd_trocr_config;
int vtd_trocr_create(const vtd_trocr_config* cfg, int max_crops, vtd_trocr** out);
void vtd_trocr_destroy(vtd_trocr* t);
/* Build options, before finalize.
 * "slots" (1 or 2, default 1): encoder-output slots.  With 2, vtd_trocr_encode_*_slot(1) may run while slot 0 still waits for or runs its
 * decode (events inside the handle order the passes); the *_slot entry points reject slot indices >= the configured count.
 * "xattn" (default 1): the form of the decoder's cross-attention (TrOCRAttention with encoder_hidden_states, reached from
 * text_recognizer.py:58).  0 -- as the reference computes it: the encoder pass projects the encoder states E to keys and values per decoder
 * layer (K = E Wk^T + bk, V = E Wv^T + bv; 28 MB per crop and slot) and every decode step reads both.  1 -- the same attention with the two
 * linear projections moved across it: scores = (q_h Wk_h) . E[t] (the q . bk term is the same for every token and leaves the softmax),
 * context = (sum_t P[t] E[t]) Wv_h^T + bv (the probabilities sum to 1); the slot holds E itself (0.89 MB per crop), a decode step reads 0.89
 * instead of 2.36 MB per live row and layer, the encoder pass has 24 projections less.  Results agree within the tolerances the goldens are
 * held to and greedy ids are identical on them (tests/test_gpu_trocr.py).  Encoders wider than 768 channels or with more than 16 decoder
 * heads keep form 0 (vtd_trocr_get_option tells which one runs after finalize); 2 -- form 1 or -1106 at finalize. */
int vtd_trocr_set_option(vtd_trocr* t, const char* name, int value);
int vtd_trocr_get_option(const vtd_trocr* t, const char* name, int* value);
/* One tensor of VisionEncoderDecoderModel.state_dict() (text_recognizer.py:42): "encoder.embeddings.*",
 * "encoder.encoder.layer.N.*", "encoder.layernorm.*", "decoder.model.decoder.*", "decoder.output_projection.weight" (optional: tied to
 * embed_tokens when absent); "encoder.pooler.*" is accepted and ignored.  float32, PyTorch memory order. */
int vtd_trocr_set_tensor(vtd_trocr* t, const char* key, const float* host_data, int64_t numel);
int vtd_trocr_finalize(vtd_trocr* t, vtd_stream stream);
/* text_recognizer.py:48-55 for every box (frame, x1, y1, x2, y2) of frames_dev ([n_frames,H,W,3] uint8 BGR): crop, BGR->RGB,
 * TrOCRProcessor (Pillow bilinear resize to image_size^2, /255, (x-0.5)/0.5), then the ViT encoder and the decoder's cross-attention
 * keys / values.  boxes_host: [ncrops][5] int32 in HOST memory (the filter tables depend on the crop sizes). */
int vtd_trocr_encode_crops(vtd_trocr* t, const uint8_t* frames_dev, int n_frames, int height, int width, const int32_t* boxes_host, int ncrops,
                           vtd_stream stream);
/* Alternative input: the pixel_values tensor the reference hands to generate(), [ncrops,3,S,S] float32 (text_recognizer.py:55). */
int vtd_trocr_encode_pixels(vtd_trocr* t, const float* pixel_values_dev, int ncrops, vtd_stream stream);
/* generate(pixel_values, max_length) (text_recognizer.py:58), greedy: ids_dev [ncrops][max_length] int32 receives
 * decoder_start_token_id, the arg-max tokens up to and including <eos>, then pad_token_id.  logits_dev, if not NULL, receives the
 * logits of every step as [ncrops][max_length-1][vtd_trocr_logits_stride()] float32.  forced_ids_dev ([ncrops][forced_len], optional):
 * teacher forcing -- these tokens are fed instead of the arg-max (tensor-level tests).  Synchronises the stream now and then (to stop
 * once every row has finished). */
int vtd_trocr_generate(vtd_trocr* t, int ncrops, int max_length, const int32_t* forced_ids_dev, int forced_len, int32_t* ids_dev, float* logits_dev,
                       vtd_stream stream);
/* Test taps as float32 on the host: "pixel_values" [n,3,S,S], "encoder" (last_hidden_state) [n,tokens,enc_hidden]. */
/* Two encoder-output slots per handle (vtd_trocr_num_slots): the encoder pass of the next crop batch (ViT + the cross-attention keys /
 * values of all decoder layers, MFMA-bound) can run on one stream into slot 1 while the previous batch is decoded out of slot 0 on
 * another (latency- / HBM-bound): `*_slot` variants of the three calls above; the plain ones are slot 0.  The handle orders the passes
 * itself with events (an encoder pass waits for the decode that last read its slot and for the previous encoder pass; a decode waits for
 * its slot's encoder pass and for the previous decode), so callers only choose streams.  generate() stops computing a row once it has
 * emitted </s> (rows leave a compact live list; the ids equal those of generate(), which pads such rows) and returns when all but the
 * last two steps have run; vtd_trocr_last_steps = decoder steps the last call enqueued. */
int vtd_trocr_num_slots(const vtd_trocr* t);
int vtd_trocr_encode_crops_slot(vtd_trocr* t, int slot, const uint8_t* frames_dev, int n_frames, int height, int width, const int32_t* boxes_host,
                                int ncrops, vtd_stream stream);
int vtd_trocr_encode_pixels_slot(vtd_trocr* t, int slot, const float* pixel_values_dev, int ncrops, vtd_stream stream);
/* encode_crops in two halves, so that one encoder pass + decode can serve the crops of SEVERAL frame batches (their frame sizes may
 * differ): stage_crops runs the processor (text_recognizer.py:49-52) for the boxes of one resident frame batch into rows
 * [row_offset, row_offset + ncrops) of the slot, encode_staged runs the ViT encoder and the cross-attention keys / values over the first
 * ncrops staged rows.  The decode's cost per step is mostly fixed, so merged recogniser batches are much cheaper per crop. */
int vtd_trocr_stage_crops_slot(vtd_trocr* t, int slot, const uint8_t* frames_dev, int n_frames, int height, int width, const int32_t* boxes_host,
                               int ncrops, int row_offset, vtd_stream stream);
int vtd_trocr_encode_staged_slot(vtd_trocr* t, int slot, int ncrops, vtd_stream stream);
int vtd_trocr_generate_slot(vtd_trocr* t, int slot, int ncrops, int max_length, const int32_t* forced_ids_dev, int forced_len, int32_t* ids_dev,
                            float* logits_dev, vtd_stream stream);
int vtd_trocr_last_steps(const vtd_trocr* t);
/* Measurement hook (bench.py roofline of the Transformer line): mode bit 0 (value 1) brackets the cross-attention launch of decoder layer 0 of every
 * step with HIP events on the decode stream; get_profile returns their summed time, the number of launches and the summed row counts the
 * launches read (exact live-row counts), then resets.  While profiling is on, generate() waits for its own completion. */
int vtd_trocr_set_profiling(vtd_trocr* t, int mode);
int vtd_trocr_get_profile(vtd_trocr* t, double* total_ms, int64_t* launches, int64_t* row_launches, vtd_stream stream);
/* Mode bit 1 (value 2, may be or-ed with 1): every dense-GEMM launch of the encoder pass (the kernel with the largest share of the
 * Transformer recogniser's time) is bracketed with HIP events on its stream; returns summed time, launches and executed FLOPs (2 M N K). */
int vtd_trocr_get_gemm_profile(vtd_trocr* t, double* total_ms, int64_t* launches, double* total_flops, vtd_stream stream);
int vtd_trocr_read_tap(vtd_trocr* t, const char* name, int ncrops, float* host_out, int64_t capacity, vtd_stream stream);
int vtd_trocr_encoder_tokens(const vtd_trocr* t);
int vtd_trocr_logits_stride(const vtd_trocr* t);
int64_t vtd_trocr_macs_per_crop(const vtd_trocr* t);
/* Kernel-selection table (see vtd_detector_set_tuning). */
int vtd_trocr_set_tuning(vtd_trocr* t, const char* table_text);
int64_t vtd_trocr_get_tuning(const vtd_trocr* t, char* buf, int64_t capacity);
int vtd_trocr_tuning_measured(const vtd_trocr* t);

#ifdef __cplusplus
}
#endif
#endif /* VTD_H */
