/* summertts_hip.h -- C ABI of libsummertts_hip.so, the MI355X (gfx950) acoustic + vocoder engine.
 *
 * This is the drop-in boundary for the hot path of huakunyang/SummerTTS: everything that
 * SynthesizerTrn::infer does AFTER the text frontend has produced phoneme ids
 * (/root/reference/src/models/SynthesizerTrn.cpp:357-400) and the model construction it depends on
 * (SynthesizerTrn.cpp:91-163).  Plain pointers and sizes only -- no C++/torch types -- so the
 * reference's C++ class (include/SynthesizerTrn.h in this repo keeps the reference's exact class
 * surface) or any other host language binds to it directly.  See INTEGRATION.md for the binding a
 * SummerTTS maintainer would add.
 *
 * Conventions: every function returns 0 on success or a negative STS_E* code; sts_last_error()
 * returns a static, thread-local description.  Buffers returned through `**` out-parameters are
 * malloc()'d and owned by the caller (free with sts_free == the reference's tts_free_data,
 * /root/reference/src/utils/utils.cpp:34-37).  An engine is not re-entrant (neither is the
 * reference instance); use one engine per host thread / per GPU.
 */
#ifndef SUMMERTTS_HIP_H_
#define SUMMERTTS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sts_engine sts_engine;

enum {
    STS_OK = 0,
    STS_EINVAL = -1,   /* bad argument (null pointer, phoneme id outside the vocabulary, n <= 0 ...) */
    STS_EMODEL = -2,   /* the blob does not parse as a SummerTTS model */
    STS_EDEVICE = -3,  /* HIP runtime failure (no gfx950 device, out of memory ...) */
    STS_ESTATE = -4    /* call sequence error (e.g. asking for a tap that was not recorded) */
};

/* Replaces SynthesizerTrn::SynthesizerTrn(float* modelData, int32_t modelSize)
 * (/root/reference/src/models/SynthesizerTrn.cpp:91-163): parses the float-stream blob (size in BYTES,
 * as ttsLoadModel returns it), repacks and uploads the weights to HIP device `device`.  The blob is
 * copied; the caller may free it afterwards (as test/main.cpp:144-145 does). */
int sts_create(const float* blob, int64_t blob_bytes, int device, sts_engine** out);
void sts_destroy(sts_engine* e);

/* Replaces SynthesizerTrn::getSpeakerNum (SynthesizerTrn.cpp:79-89): 1 for single-speaker models. */
int sts_speaker_num(const sts_engine* e);

/* Model facts the host side needs (vocabulary size for id validation, samples per frame, ...). */
typedef struct sts_model_info {
    int32_t is_multi_speaker, lang_type, dur_pred_type, dec_type;
    int32_t vocab, hidden, inter_channels, speaker_num, gin_channels;
    int32_t samples_per_frame;      /* total upsampling factor */
    int32_t sample_rate;            /* 16000 (/root/reference/test/main.cpp:13,16) */
    int64_t blob_floats_consumed;   /* floats consumed by the acoustic sections (frontend sections follow) */
} sts_model_info;
int sts_get_info(const sts_engine* e, sts_model_info* info);

/* Replaces the post-frontend part of SynthesizerTrn::infer (SynthesizerTrn.cpp:357-400) for one
 * utterance: ids[n] -> int16 PCM.  Out-of-range sid -> 0 (SynthesizerTrn.cpp:366-369).
 * *pcm_out is malloc()'d; *n_out = sample count. */
int sts_infer_ids(sts_engine* e, const int32_t* ids, int32_t n, int32_t sid, float length_scale,
                  int16_t** pcm_out, int32_t* n_out);

/* Streaming form (no reference counterpart: SynthesizerTrn::infer returns the whole utterance,
 * SynthesizerTrn.cpp:389-400).  Text encoder, duration predictor and flow run once; the decoder then runs
 * chunk by chunk (chunk_frames acoustic frames each, decoded with the receptive-field halo on both sides) and
 * `cb` receives every chunk's PCM as soon as it is on the host: first audio after one chunk instead of after
 * the whole utterance, decoder workspace bounded by the chunk size.  The concatenated chunks equal
 * sts_infer_ids' output bit for bit when the kernel variant is pinned (sts_set_conv_mode) and to within 1 LSB
 * under the automatic choice (a chunk is a smaller launch and may be routed to the split-K kernel, which sums
 * K in a different order).  `pcm` is only valid during the callback; a non-zero return stops the stream.
 * *n_total = samples delivered. */
typedef int (*sts_chunk_cb)(void* user, const int16_t* pcm, int32_t n_samples, int32_t sample_offset);
int sts_infer_ids_stream(sts_engine* e, const int32_t* ids, int32_t n, int32_t sid, float length_scale,
                         int32_t chunk_frames, sts_chunk_cb cb, void* user, int32_t* n_total);
/* frames of context the streaming decoder adds on each side of a chunk (a property of the loaded model) */
int sts_stream_halo_frames(const sts_engine* e);

/* Batched form (new capability; the reference processes exactly one utterance per call).  The B
 * utterances are packed along time on the device and run through every kernel together.
 * pcm_out[b] is malloc()'d per utterance. */
int sts_infer_ids_batch(sts_engine* e, int32_t B, const int32_t* const* ids, const int32_t* n,
                        const int32_t* sid, const float* length_scale, int16_t** pcm_out, int32_t* n_out);

/* Same, but leaves the PCM on the device (packed back to back, utterance order) for a following
 * RCCL gather; n_out[b] = samples of utterance b, *total_out = sum.  Copy out with
 * sts_copy_pcm_device (device-to-device, asynchronous on the engine stream then synchronised). */
int sts_run_batch(sts_engine* e, int32_t B, const int32_t* const* ids, const int32_t* n, const int32_t* sid,
                  const float* length_scale, int32_t* n_out, int64_t* total_out);
/* sts_set_host_pcm(e, 1): sts_run_batch also downloads the PCM into an engine-owned pinned host buffer as the last step of
 * the run (one stream synchronisation per call instead of two); sts_copy_pcm_host then copies from that buffer.  Leave it
 * off (default) when the PCM stays on the device for an RCCL gather. */
int sts_set_host_pcm(sts_engine* e, int enable);
int sts_copy_pcm_device(sts_engine* e, void* device_dst, int64_t capacity_samples);
int sts_copy_pcm_host(sts_engine* e, int16_t* host_dst, int64_t capacity_samples);
/* Zero-copy form of sts_copy_pcm_host for a caller that consumes the PCM before its next call on this engine: *pcm points into
 * the engine-owned pinned host buffer the last run downloaded into (needs sts_set_host_pcm(e, 1); STS_ESTATE otherwise), *count =
 * samples of all utterances back to back.  The pointer is valid until the next run / destroy on this engine. */
int sts_pcm_host_view(sts_engine* e, const int16_t** pcm, int64_t* count);

/* Parity/diagnostic controls (test infrastructure hooks; defaults reproduce the reference):
 *   forced durations: per-phoneme frame counts that override ceil(exp(logw)*lengthScale) for the NEXT
 *   run only (the ceil is discontinuous, so waveform parity is checked with the
 *   oracle's durations).  Packed for the whole batch (sum of n entries). */
int sts_set_forced_durations(sts_engine* e, const int32_t* dur, int64_t count);
/*   record intermediate tensors of the next run: "x_enc","m","logw","z_p","z","wave" */
int sts_set_record_taps(sts_engine* e, int enable);
/*   fetch a tap: malloc()'d copy, channel-major [channels][total_len] (== the reference's column-major
 *   MatrixXf [time, channels]); for batches the utterances are packed along time. */
int sts_get_tap(sts_engine* e, const char* name, float** data, int32_t* channels, int64_t* length);
/*   durations of the last run, packed (sum of n entries) */
int sts_get_durations(sts_engine* e, int32_t* dur, int64_t capacity);
/*   conv dispatch: 0 = automatic, 1 = force the generic VALU kernel everywhere, 2..7 = force LDS-staged
 *   matrix-core tile (idx-2), 8 / 9 = force the split-K matrix-core kernel (32 / 64 columns per workgroup) */
int sts_set_conv_mode(sts_engine* e, int mode);
/*   arithmetic of the decoder trunk's matrix-core convs (upsamplers + ResBlock convs, ~95 % of the FLOPs):
 *   0 = fp32 operands split exactly into three bf16 terms each, six bf16 MFMA products per fp32 product, fp32 accumulation
 *       (conv_bf3.hip; as accurate as 1 against float64, 6/16 of its matrix-pipe time);
 *   1 = the exact-fp32 MFMA instruction (v_mfma_f32_32x32x2_f32) everywhere;
 *   2 = as 0, and also for every other eligible matrix-core conv (flow, text encoder) regardless of its grid size -- by default
 *       those switch to the split form only from the batch size on at which they stop being launch-latency-bound (tests).
 *   3 = "f16x2": fp32 operands as TWO fp16 terms (the small one pre-scaled by 2^11, weights by a per-conv power of two), three
 *       fp16 MFMA products per fp32 product -- half the matrix-pipe time of 0, 22-23 instead of 24 operand bits (measured error
 *       against float64: docs/HISTORY.md 5f) -- the default.  An activation beyond fp16's range raises a flag and the call (a
 *       streaming call: the chunk, before it is handed out) is repeated in form 0; sts_profile.conv_math_fallbacks counts these.
 *       After two such calls in a row the engine stays in form 0 until sts_set_conv_math is called again
 *       (sts_profile.conv_math_pinned).  The first repeat of an engine and the pinning are reported through tts_log.
 *   The default can also be chosen with the environment variable STS_CONV_MATH = f16x2 | bf16x3 | f32; any other value is
 *   reported through tts_log and ignored (the default applies). */
int sts_set_conv_math(sts_engine* e, int mode);

/*   test hooks (per engine, never read from the environment): force a kernel family that the automatic choice would not pick
 *   at the size of a test.  key: STS_DBG_ATTN_BLOCK_MIN_WGS -- the matrix-core block attention kernel engages from this many
 *   workgroups on (default 96; 1 = always).
 *   (Keys 2-4 selected the two persistent-kernel families of round 3; both lost their A/B against the launch path and were deleted
 *   in round 5 -- the numbers stay retired and answer STS_EINVAL.) */
enum { STS_DBG_ATTN_BLOCK_MIN_WGS = 1,
       STS_DBG_TAIL_FUSED = 14 /* MB-iSTFT / MS-iSTFT decoders: 1 (default) the tail (spectrum, inverse DFT + overlap-add, synthesis filter, int16 cast) as one launch, 0 three */,
       STS_DBG_UPS_ROWPH = 15 /* upsamplers (transposed convs, stride 2 / 4 / 8): 1 (default) phases interleaved along the packed rows -> whole-sector stores, 0 phase-major rows */,
       STS_DBG_CHAIN_STREAMS = 13 /* lab: bit i = the ResBlock chains of decoder stage i as per-chain launches on three prioritised streams instead of one grouped launch per layer (-1: off) */,
       STS_DBG_H2P = 11 /* decoder stages of 128 k channels under the two-term fp16 arithmetic: 1 (default) pre-split channel-minor activations (conv_h2p.hip) from ~8 tiles of 128 x 128 per CU on, 2 always (tests), 0 the staged kernels */,
       STS_DBG_H2P_TILE = 12 /* lab: tile code of conv_h2p_group, -1 automatic */,
       STS_DBG_MEMO_CLEAR = 10 /* any value: forget the launch-ahead memo (bench.py: every timed request is then one the engine has not served before) */,
       STS_DBG_PCM_DIRECT = 9 /* sts_set_host_pcm(1), one utterance: 1 (default) the decoder's last kernel writes the PCM into the pinned host buffer itself, 0 a download behind it */,
       STS_DBG_DDS_TAIL = 8 /* stochastic duration predictor: 1 (default) a ConvFlow's projection + spline step ride in its last DDSConv layer's launch, 0 three launches */,
       STS_DBG_ATTN_REG = 7 /* one-query attention: 1 (default) operands in registers (attention_reg_kernel), 0 the round-1 kernel */,
       STS_DBG_LAUNCH_AHEAD = 6 /* one-utterance calls and packed batches: 1 (default) a request the engine has served before (same ids, speaker, length scale: the frame count is a pure function of them) enqueues flow + decoder before the count reaches the host, 0 the host always waits for it, 2 (tests) the memo is keyed by the phoneme count alone -- provokes the repeat that answers a hash collision */,
       STS_DBG_FLOW_FUSED = 5 /* reverse flow: 1 (default) one launch per WaveNet layer (wn_flow.hip, under the two-term fp16 arithmetic), 0 one launch per conv */ };
int sts_debug_set(sts_engine* e, int key, int value);

/* Per-stage device timing of the last run, measured with HIP events on the engine's own stream. */
typedef struct sts_profile {
    float ms_text_encoder, ms_duration, ms_flow, ms_decoder, ms_total_device;
    float ms_decoder_mfma;          /* time of the pure conv_mfma launch sequence inside the decoder */
    int32_t decoder_mfma_launches;
    double flops_text_encoder, flops_duration, flops_flow, flops_decoder;   /* algorithmic (true-tap) FLOPs */
    double flops_decoder_mfma;      /* FLOPs executed by the launches timed in ms_decoder_mfma */
    double bytes_decoder_min;       /* algorithmic HBM bytes of the decoder (weights once + act in/out per conv) */
    int64_t frames, samples, phonemes;
    double flops_decoder_mfma_executed;   /* matrix-core FLOPs the timed launches actually execute: the Winograd-domain layer
                                             kernels need (4 n3 + 3 n2) / (2 k) of a k-tap conv's direct-form products */
    double bytes_text_encoder, bytes_duration, bytes_flow;   /* algorithmic HBM bytes per stage (as bytes_decoder_min) */
    float ms_sync_wait_host;        /* host time blocked on the frame-count download (the one data-dependent sync) */
    double flops_decoder_bf16_issued;     /* bf16 matrix-core FLOPs issued by the timed launches that run on split operands
                                             (6 x their algorithmic FLOPs; 3 x with sts_set_conv_math(3)); 0 with sts_set_conv_math(1) */
    int64_t conv_math_fallbacks;          /* sts_set_conv_math(3): calls of this engine so far that were repeated in the split-bf16 form */
    int32_t conv_math_pinned;             /* 1: after two such calls in a row the engine now stays in the split-bf16 form (until sts_set_conv_math) */
    int32_t launch_ahead;                 /* 1: this run enqueued flow + decoder before the frame count reached the host (one utterance or a packed batch whose members the engine has all served before; ms_sync_wait_host ~ 0) */
    int64_t launch_ahead_misses;          /* launch-ahead runs of this engine so far (one utterance or a batch) whose remembered frame counts turned out wrong -- a hash collision of the memo -- so that flow + decoder were repeated the waiting way */
    float us_host_setup;                  /* host time from the entry of the run to the first launch being enqueued (input checks, tables, the one upload) */
    float us_host_enqueue;                /* host time from the entry of the run to the last launch being enqueued (the GPU runs behind it) */
    float us_host_tail;                   /* host time from the return of the run's last stream synchronisation to the return of the call */
} sts_profile;
/* enable: 0 off; 1 HIP events at all eight stage boundaries of a run (ms_text_encoder ... ms_decoder_mfma); 2 only the two events around the
 * decoder's matrix-core region (ms_decoder_mfma; the per-stage times read 0) -- every event is a barrier packet between two kernels
 * (5-10 us of idle GPU each), so a caller who wants the dominant kernel family timed inside a step it also times uses 2. */
int sts_set_profiling(sts_engine* e, int enable);
/* The struct only ever grows at its end (STS_ABI_VERSION counts the revisions).  sts_get_profile_ex copies min(size_bytes,
 * sizeof(sts_profile)) bytes, so a client compiled against an older header passes ITS sizeof and is never overrun;
 * sts_get_profile(e, p) == sts_get_profile_ex(e, p, sizeof(sts_profile)) of the header this library was built from -- use it only
 * when client and library are built together. */
#define STS_ABI_VERSION 7
int sts_abi_version(void);
/* bit 0: lab build (-DSTS_EXPERIMENTS: environment knobs of knobs.hpp, every conv tile code);
 * 0 for the shipped library */
int sts_build_flags(void);
int sts_get_profile(const sts_engine* e, sts_profile* p);
int sts_get_profile_ex(const sts_engine* e, void* p, int64_t size_bytes);

/* Stand-alone conv entry for op-level parity tests: y = conv1d(x) with x [Cin][L] on the host.
 * w is the reference layout [out][k][in] (transposed: same).  mode as sts_set_conv_mode; 13 / 20.. = the split-bf16 kernel
 * (automatic tile / tile code mode - 20), 50 / 60.. = the two-term fp16 kernel (automatic tile / tile code mode - 60). */
int sts_debug_conv1d(int device, const float* x, int32_t Cin, int32_t L, const float* w, const float* bias,
                     int32_t Cout, int32_t k, int32_t pad, int32_t dil, int32_t stride_transposed,
                     int32_t depthwise, float in_slope, int32_t in_act, int mode, float** y, int32_t* Lout);

/* Same conv, additionally timed: `iters` back-to-back launches between two HIP events; *ms_out = mean
 * milliseconds per launch (kernel micro-benchmarks, tools/conv_bench.py). */
int sts_debug_conv1d_bench(int device, const float* x, int32_t Cin, int32_t L, const float* w, const float* bias,
                           int32_t Cout, int32_t k, int32_t pad, int32_t dil, int32_t stride_transposed,
                           int32_t depthwise, float in_slope, int32_t in_act, int mode, float** y, int32_t* Lout,
                           int32_t iters, float* ms_out);

/* One "same"-padded conv (odd k, pad = dil (k - 1) / 2) through the pre-split path of the wide decoder stages (conv_h2p.hip): x fp32
 * [Cin][L] -> split_planes(in_slope) -> conv_h2p_group with `members` identical members -> member 0's three output forms, each decoded to
 * fp32 [Cout][L] (null: not wanted): y = the channel-major fp32 output, y16 = the channel-minor fp32 copy, yp = the two fp16 planes of
 * lrelu(out, out_slope) recombined.  res: optional residual [Cout][L].  tile < 0: automatic. */
int sts_debug_conv_h2p(int device, const float* x, int32_t Cin, int32_t L, const float* w, const float* bias, int32_t Cout, int32_t k,
                       int32_t dil, const float* res, float in_slope, float out_slope, int tile, int members, float* y, float* y16,
                       float* yp, int32_t iters, float* ms_out);

/* The same conv through the Winograd-domain lab kernel (conv_h2w.hip: segmented F(2,3) / F(2,2) on two-term fp16 operands): y = fp32 [C][L],
 * y16 = its channel-minor output of lrelu(out, out_slope) decoded to [C][L].  C % 128 == 0, odd k >= 3, (k - 1) dil <= 64. */
int sts_debug_conv_h2w(int device, const float* x, int32_t C, int32_t L, const float* w, const float* bias, int32_t k, int32_t dil, const float* res,
                       float in_slope, float out_slope, int members, float* y, float* y16, int32_t iters, float* ms_out);

void sts_free(void* p);
const char* sts_last_error(void);

/* Host-only diagnostic: the Winograd-domain weight transform the loader applies to the decoder's ResBlock convs
 * (segmented F(2,3), summertts_amd/csrc/kernels.hpp: wino_pack).  w is [Cout][k][Cin] (the blob's order); out must
 * hold n_seg * 4 * Cin_pad * Cout_pad floats, laid out [seg][4][Cin_pad][Cout_pad] (Cin_pad = Cin rounded up to 16,
 * Cout_pad to 32).  Returns n_seg (>= 1) or a negative STS_E* code.  No GPU needed. */
int sts_debug_wino_pack(const float* w, int32_t Cout, int32_t k, int32_t Cin, float* out, int64_t out_floats);

/* ---- request pool (no reference counterpart: SynthesizerTrn::infer is one blocking call per
 * utterance, SynthesizerTrn.cpp:323).  n_engines engines on one GPU, one worker thread each, one FIFO; a free
 * worker folds up to max_batch queued requests into ONE packed variable-length batch.  submit() returns a
 * ticket (> 0) or a negative STS_E* code; wait() blocks until that request is done and hands back a
 * malloc()'d PCM buffer (release with sts_free).  Thread-safe: any thread may submit / wait. */
typedef struct sts_pool sts_pool;
int sts_pool_create(const float* blob, int64_t blob_bytes, int device, int n_engines, int max_batch, sts_pool** out);
void sts_pool_destroy(sts_pool* p);
int64_t sts_pool_submit(sts_pool* p, const int32_t* ids, int32_t n, int32_t sid, float length_scale);
int sts_pool_wait(sts_pool* p, int64_t ticket, int16_t** pcm_out, int32_t* n_out);
int sts_pool_stats(sts_pool* p, int64_t* batches, int64_t* requests);
const char* sts_pool_last_error(void);

/* ---- multi-device batch (no reference counterpart).  One host process drives n_devices GPUs:
 * one engine (weights replicated) and one worker thread per entry of `devices` (HIP device indices; an index may repeat,
 * e.g. {0, 0} = two engines on one GPU).  sts_multi_infer_ids_batch shards the B utterances by utterance -- longest first,
 * greedy, balanced by phoneme count, no exchange between devices -- runs every shard as one packed batch on its device
 * concurrently, and returns the PCM in INPUT order: pcm_out[b] is malloc()'d per utterance (release with sts_free),
 * n_out[b] = its sample count.  On any failure every output is released and a negative STS_E* code is returned.
 * The handle is not re-entrant (one batch at a time), like an engine. */
typedef struct sts_multi sts_multi;
int sts_multi_create(const float* blob, int64_t blob_bytes, const int32_t* devices, int32_t n_devices, sts_multi** out);
/*   How the PCM comes home.  STS_MULTI_AUTO (= sts_multi_create) and STS_MULTI_DOWNLOAD: every device downloads its own shard over
 *   PCIe.  STS_MULTI_RCCL (opt-in): one RCCL communicator rank per device (ncclCommInitAll; distinct devices required, one is
 *   allowed): sample counts and a "rank 0 can receive" word by ncclAllGather, the int16 PCM of ranks > 0 by ncclSend / grouped
 *   ncclRecv into a gather buffer on devices[0] (over xGMI), then ONE download.  A failure or a 60 s timeout inside a collective
 *   aborts all communicators of the handle (ncclCommAbort): that call returns STS_EDEVICE -- it never hangs -- and the handle
 *   continues with per-device downloads.  The RCCL path is not part of the automatic mode because it has not been run on N > 1 real
 *   devices yet.  sts_multi_gather_mode reports which one a handle uses (1 = RCCL). */
enum { STS_MULTI_AUTO = 0, STS_MULTI_RCCL = 1, STS_MULTI_DOWNLOAD = 2 };
int sts_multi_create_ex(const float* blob, int64_t blob_bytes, const int32_t* devices, int32_t n_devices, int32_t flags, sts_multi** out);
int sts_multi_gather_mode(const sts_multi* m);
/*   sts_multi_rccl_ranks: the size of the handle's communicator as RCCL reports it (ncclCommCount; 0 = no RCCL gather on this handle,
 *   -1 = the RCCL library has no ncclCommCount).  sts_multi_last_gather_ms: wall time rank 0 spent inside the last call's RCCL gather
 *   (count exchange, ready round, transfers, the one download; 0 in download mode).  sts_multi_set_conv_math: sts_set_conv_math on
 *   every engine of the handle. */
int sts_multi_rccl_ranks(sts_multi* m);
double sts_multi_last_gather_ms(const sts_multi* m);
int sts_multi_set_conv_math(sts_multi* m, int mode);
/*   test hook: the shared library that provides the nccl* entry points (NULL / "" = librccl.so.1) and whether STS_MULTI_RCCL may list
 *   one device several times (tests/fake_rccl: N emulated ranks on one GPU; real RCCL refuses duplicates).  Only before the first
 *   STS_MULTI_RCCL handle of the process is created.  TEST-ONLY: refused with STS_ESTATE unless the process environment carries
 *   STS_TEST_HOOKS=1, so that no caller of the drop-in library substitutes the collective library or lifts the distinct-device check
 *   by accident. */
int sts_multi_set_rccl_library(const char* path, int allow_repeated_devices);
/*   layout of the gather buffer (host arithmetic only): rank r's block starts at offsets[r] samples (256-byte aligned);
 *   returns the buffer's extent in samples */
int64_t sts_multi_gather_layout(const int64_t* counts, int32_t n_ranks, int64_t* offsets);
void sts_multi_destroy(sts_multi* m);
int sts_multi_device_count(const sts_multi* m);
int sts_multi_speaker_num(const sts_multi* m);
int sts_multi_infer_ids_batch(sts_multi* m, int32_t B, const int32_t* const* ids, const int32_t* n, const int32_t* sid,
                              const float* length_scale, int16_t** pcm_out, int32_t* n_out);
/* the placement sts_multi_infer_ids_batch would use: device_slot_out[b] = index into `devices` (host logic only) */
int sts_multi_shard_of(const sts_multi* m, int32_t B, const int32_t* n, int32_t* device_slot_out);
const char* sts_multi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
