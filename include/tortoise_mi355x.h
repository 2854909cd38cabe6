/*
 * tortoise_mi355x.h — C ABI of the MI355X-native Tortoise-TTS hot path.
 *
 * The reference (balisujohn/tortoise.cpp) exports no library API: its three stage drivers talk to
 * the tensor runtime through the ggml backend seam — named graph inputs set with
 * ggml_backend_tensor_set, one blocking ggml_backend_graph_compute, the result fetched with
 * ggml_backend_tensor_get. Each entry point below replaces one such
 * {set inputs, compute, get output} group; the reference call sites are cited per function
 * (file:line in /root/reference). INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: plain C types only; the caller owns every host buffer; all calls are synchronous
 * (results are valid on return); no exceptions cross the boundary — functions return TTS_OK (0)
 * or a negative tts_status and tts_last_error() describes the failure. One tts_ctx per GPU and
 * per host thread (the reference is single-threaded with global state, main.cpp:47-50).
 *
 * Layouts follow the reference's host vectors: logits [B][8194]; latents [rows][1024];
 * x_t / mel [100][T] (time fastest); network output [200][T]; vocoder noise [64][T+10]; audio
 * [(T+10)*256-6].
 */
#ifndef TORTOISE_MI355X_H
#define TORTOISE_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tts_ctx tts_ctx;

typedef enum {
  TTS_OK = 0,
  TTS_ERR_ARG = -1,    /* bad argument / call order */
  TTS_ERR_IO = -2,     /* cannot open / truncated file */
  TTS_ERR_FORMAT = -3, /* bad magic, unknown tensor name, wrong shape (main.cpp:834-873) */
  TTS_ERR_HIP = -4,    /* HIP runtime failure (no device, OOM, launch error) */
  TTS_ERR_STATE = -5,  /* stage not loaded / begin() not called */
  TTS_ERR_LIMIT = -6   /* exceeds a reference limit (404 text/608 mel positions, 500 codes) */
} tts_status;

enum { TTS_VOCAB_MEL = 8194, TTS_DMODEL = 1024, TTS_MEL_CH = 100, TTS_CODES = 502 };

/* ---- lifecycle --------------------------------------------------------------------------- */
/* Interface version: bumped whenever a prototype in this file changes incompatibly (a caller built against another value must not call in).
 *   4 = round 4: tts_host_mel_diffusion100 gained `normalize` (voice files written by the earlier tools/make_voice.py hold a NORMALISED mel of
 *       full-length clips and must be regenerated: INTEGRATION.md "voice files"),
 *   6 = round 6: options "latency_mode", "fp16_check"; tts_diffusion_fp16_check, tts_device_numa_node, tts_pin_to_device_numa_node; the split proj_out weight is
 *       scaled per tensor
 *   5 = round 5: tts_ar_set_stop_schedule, tts_version; option "attn_proj_f16"; the default AttentionBlock multiplies proj_out on an F32-accurate
 *       (split fp16 pair) weight. */
#define TTS_API_VERSION 6
int tts_version(void);

/* replaces ggml_backend_cuda_init(0) (main.cpp:651, 1213, 1777). device = HIP ordinal; returns NULL
 * when there is no such device. device = -1 gives a host-only context (tokenizer, RNG, sampler):
 * every stage call on it fails with TTS_ERR_HIP — there is no CPU compute path. */
tts_ctx *tts_create(int device);
/* Host placement of a device context (round 6; multi-GPU runs: one process per GPU, each with a pool of sampler threads): the NUMA node of the context's GPU
 * (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node), -1 if unknown or a host-only context; cpulist_out receives that node's CPU list in the kernel's
 * "0-31,128-159" form (empty when the node is unknown). tts_pin_to_device_numa_node restricts the calling thread — and the sampler threads it creates later — to it
 * (sched_setaffinity) and returns the number of CPUs in the mask, 0 if nothing was changed. The reference has no counterpart (single device, main.cpp:651). */
int tts_device_numa_node(const tts_ctx *ctx, char *cpulist_out, int cpulist_cap);
int tts_pin_to_device_numa_node(tts_ctx *ctx);
void tts_destroy(tts_ctx *ctx);
const char *tts_last_error(const tts_ctx *ctx);
/* Options (all have reference defaults): "gn_eps" (1e-6; ggml's GroupNorm epsilon),
 * "ggml_lut" (0/1: emulate ggml-CPU fp16 lookup tables for GELU/SiLU),
 * "prof_only:<family>" (1: add the family to the list of profiled families, 0: clear the list = all families),
 * "prof_stride" (1 default: every launch of a profiled family is bracketed by an event pair; N: every Nth launch —
 * an event pair drains the pipeline around the launch, so bracketing all 9 600 GEMM launches of a pass costs ~5 %),
 * "sampler_threads" (-1 default = min(7, hardware threads - 1); 0 = sample on the calling thread; the token ids
 * do not depend on it: the RNG is consumed in candidate order before the per-candidate scans run),
 * "share_uncond" (1 default: in tts_diffusion the conditioning_timestep_integrator layers of the unconditioned branch,
 * whose input does not depend on the candidate, are evaluated once per distinct sequence length instead of once per
 * candidate; 0 = once per candidate. Same arithmetic per row either way),
 * "ar_weights" (0 default: the decode step streams the f32 weights, reference numerics; 1 — set BEFORE tts_load_ar — the decode
 * step streams fp16 copies (half the bytes; logits ~1e-3 off, so sampled ids diverge from the f32 mode after some steps: the
 * throughput mode of SURVEY 8d; prefill and latent pass stay f32-exact); 2 — also BEFORE tts_load_ar — OCP fp8 (e4m3) copies with one
 * power-of-two scale per output column (a quarter of the bytes, logits ~5e-2 off: SURVEY 8 f4),
 * "diff_graph" (1 default: tts_diffusion captures ONE sampling step — ~125 kernels, every per-step value read through a device-side step
 * counter — into a hipGraph and replays it; 0: every step is launched kernel by kernel), "prof_eager_every" (8: while a diff_* family is
 * being profiled every 8th step runs eagerly with its event pairs, the rest replay the graph),
 * "rng_shard_offset" / "rng_shard_total" (0 / 0 default = unsharded): candidate-parallel multi-GPU runs. This context
 * holds candidates [offset, offset + n_candidates) of a batch of `total`: the sampler skips the uniforms of the other
 * ranks' candidates (the used uniform of (step s, candidate c) is output 2 (s total + c) + 1 of the mt19937 stream), and
 * TTS_NOISE_DEVICE streams are keyed by the global candidate id — so G ranks x B/G candidates reproduce one rank x B,
 * "stream_cus" (0 default = whole chip; set BEFORE any model is loaded): n > 0 puts this context's stream on the n lowest CUs of every
 * XCD, n < 0 on all but those (hipExtStreamCreateWithCUMask), so that two contexts of one process can split the GPU. Results do not
 * depend on it. Measured use: profiles/r3_stage_overlap_probe.txt (the AR stage does not tolerate a partition; kept as a tool).
 * "attn_f32" (0 default): 1 = the diffusion stage's AttentionBlock in REFERENCE PRECISION. The reference evaluates QK^T, softmax, PV and proj_out as F32
 * ggml_mul_mat / ggml_soft_max (main.cpp:3848-3875); the default feeds them to the matrix cores as fp16 operands (the throughput mode). With 1 every one of
 * those products runs on split-precision fp16 pairs (x = hi + lo, three MFMAs per product, 2^-22 relative) and SiLU is the reference's f32 formula: the
 * 80-step sampling loop then stays as close to the CPU restatement as a second f32 evaluation of the reference's graph does (tests/golden/parity_floor.json).
 * Costs about 1.5x the diffusion stage's time; may be switched between calls.
 * "lc_attn_f32" (1 default): the latent conditioner's AttentionBlocks (main.cpp:3156-3321; evaluated once per utterance, their output enters every sampling step) run in
 * the reference precision of "attn_f32" whatever that option says — an fp16 rounding inside them would be the same perturbation at all 80 steps; 0 = follow "attn_f32".
 * With the defaults (attn_f32 0, attn_proj_f16 0, lc_attn_f32 1) the 80-step loop sits on the same floor as attn_f32 = 1 (1.6e-3 max / 7.0e-5 mean at the benchmark's
 * length against 1.3e-3 / 6.1e-5) at +4 % of the stage's time instead of +46 %; the parity tests hold both to the same gates.
 * "attn_proj_f16" (0 default): the default (throughput) AttentionBlock keeps q, k, v, the softmax numerators and the attention output as fp16 MFMA
 * operands but multiplies proj_out — an F32 linear in the reference — on its weight held as the split pair W_hi + W_lo (two MFMAs per product). Of the five
 * fp16 roundings of the rounds 1-4 block only the WEIGHT's survives 80 steps (the same perturbation at every step; tests/golden/parity_floor.json
 * "ablation"): without it the default mode sits on the f32-vs-f32 floor of the sampling loop. 1 = the all-fp16 block of rounds 1-4 (A/B only).
 * "device_topk" (1 default): tts_autoregressive's decode loop samples from the device prefilter's lists (tts_ar_step_sample); 0 = every step copies
 * the [B][8194] logits to the host as the reference does (main.cpp:4766-4768). Sampled ids are identical either way.
 * "dec_f32_mfma" (0 default; set BEFORE tts_load_ar): the decode step's LayerNorm-GEMV kernels multiply on v_mfma_f32_16x16x4_f32 (exact f32 products)
 * instead of split-precision fp16 pairs.
 * "latency_mode" (0 default; round 6): small diffusion batches only (at most 2 048 packed rows: one utterance with both guidance branches — the reference's own
 * workload is ONE, main.cpp:6570). The GroupNorm statistics of every f32 tensor of the sampling step are accumulated by the epilogue of the GEMM that produces it
 * (exact fixed-point sums per sequence and 32-channel group) and the 45 GroupNorms of a step become elementwise launches. Results are reproducible run to run and
 * independent of the other candidates of the (small) batch, held to the same oracle gates as the default, but NOT bit-identical to the default path, which is why it is
 * opt-in. Measured gain: 2-5 % of the single-utterance diffusion stage (138.7 -> 135.7 ms in the bench line, profiles/r6_small_batch.txt); it loses above one utterance.
 * "hoist_integrator" (1 default), "attn_q64" (0 default): INTEGRATION.md; both bit-identical to the setting they replace.
 * "fp16_check" (0 default): see tts_diffusion_fp16_check.
 * "load_threads" (0 default = min(16, hardware threads); set BEFORE tts_load_*): host threads that read, upload and (diffusion) re-lay-out the tensors; 1 = serial.
 * "load_device_pack" (1 default; set BEFORE tts_load_ar): decode layouts built by kernels from the uploaded file tensors (0: by the host threads). Same bits.
 * "noise_pipeline", "rng_fast_normal" (1 default): production of TTS_NOISE_REFERENCE draws (beside the device loop; two-phase normal distribution). Results and the RNG
 * state afterwards are those of single std::normal_distribution draws either way (tests/test_host_parity.py); 0 = the single-draw forms. */
int tts_set_option(tts_ctx *ctx, const char *key, double value);

/* ---- weight files (drop-in format: magic 0x67676d6c + name-keyed F32 records) ------------- */
/* autoregressive_model_load, main.cpp:482-897 */
int tts_load_ar(tts_ctx *ctx, const char *path);
/* diffusion_model_load, main.cpp:931-1634 */
int tts_load_diffusion(tts_ctx *ctx, const char *path);
/* vocoder_model_load, main.cpp:1665-2021 */
int tts_load_vocoder(tts_ctx *ctx, const char *path);
/* number of transformer / main diffusion layers found in the file (30 / 10 for real weights) */
/* CLVP candidate re-ranker. NOT in the reference, which writes candidate 0 (main.cpp:6575): upstream tortoise-tts
 * scores every candidate's codes against the text with CLVP (tortoise/models/clvp.py, use_xformers=True) and keeps the best.
 * File: the reference's container format, tensor names of the upstream state dict (tortoise.cpp_amd/synth_weights.py: write_clvp). */
int tts_load_clvp(tts_ctx *ctx, const char *path);
/* Voice-conditioning encoder. NOT in the reference, which reads the finished 1024-float latent from --voice
 * (main.cpp:5179-5184; README.md:54-72 is an offline PyTorch recipe): upstream tortoise-tts' UnifiedVoice.get_conditioning =
 * ConditioningEncoder(80 mel bands -> 1024, 6 attention blocks, 16 heads), position 0 of every clip, mean over the clips.
 * File: the reference's container format with the upstream state dict's `conditioning_encoder.*` tensors (tools/convert_weights.py
 * --conditioning-encoder). tts_voice_latent: mel = the clips' 80-band log-mel spectrograms [80][frames[c]] one after the other (the audio
 * front-end — STFT, mel filterbank, normalisation — stays with the caller); out1024 = what a --voice file holds. */
int tts_load_voice_encoder(tts_ctx *ctx, const char *path);
/* The other voice latent: upstream DiffusionTts.get_conditioning (contextual_embedder: two k = 3 / stride 2 convolutions, five 2048-channel
 * attention blocks with relative position bias, mean over the frames of all clips) turns the clips' 100-band mel [100][frames[c]] into the
 * 2048 floats the reference reads as the WEIGHT `diffusion_conditioning_latent` of ggml-diffusion-model.bin (main.cpp:1557-1560: one voice
 * per weight file). tts_set_diffusion_conditioning_latent replaces that weight in the loaded diffusion model (after tts_load_diffusion). */
int tts_load_diffusion_conditioning_encoder(tts_ctx *ctx, const char *path);
int tts_diffusion_conditioning_latent(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_clips, float *out2048);
int tts_set_diffusion_conditioning_latent(tts_ctx *ctx, const float *latent2048);
int tts_voice_latent(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_clips, float *out1024);
int tts_ar_layers(const tts_ctx *ctx);
int tts_diffusion_layers(const tts_ctx *ctx);

/* ---- RNG (main.cpp:47-50, 6546): std::mt19937 + uniform<float> + normal<double> ------------ */
void tts_seed(tts_ctx *ctx, uint32_t seed);
/* libstdc++ text state ("fin >> generator", main.cpp:6260-6262, 6475-6477) */
int tts_rng_load_state(tts_ctx *ctx, const char *path);
/* "fout << generator": hands the engine state back to a host program that keeps its own std::mt19937 */
int tts_rng_save_state(tts_ctx *ctx, const char *path);
float tts_rng_uniform(tts_ctx *ctx);
void tts_rng_normal(tts_ctx *ctx, float *out, int64_t n);

/* ---- host front-end (common.cpp:166-339, main.cpp:6559-6567) ------------------------------- */
int tts_tokenizer_load(tts_ctx *ctx, const char *tokenizer_json);
/* " " -> "[SPACE]", greedy longest match, wrapped with 255 ... 0. Returns the id count. */
int tts_tokenize(tts_ctx *ctx, const char *message, int32_t *ids_out, int cap);

/* ---- autoregressive stage ------------------------------------------------------------------ */
/* Sets the graph inputs of the prefill (input_tokens, input_position, auto_conditioning;
 * main.cpp:5136-5184) and sizes the per-candidate KV cache for P + max_steps positions
 * (the reference: fixed 404 x batch 4, main.cpp:794-797). */
int tts_ar_begin(tts_ctx *ctx, const int32_t *text_ids, int n_text, const float *voice1024,
                 int n_candidates, int max_steps);
/* autoregressive_graph(fake_inputs=true) + compute + tensor_get("next token logits")
 * (main.cpp:5131-5186, 4766-4768). logits_out: [B][8194] host floats. */
int tts_ar_prefill(tts_ctx *ctx, float *logits_out);
/* autoregressive_graph(false, n_past=P+i, fixed_position=i+2) + compute (main.cpp:5227-5247):
 * prev_ids[B] = input_mel_tokens, step index i. logits_out as above. */
int tts_ar_step(tts_ctx *ctx, const int32_t *prev_ids, int step_i, float *logits_out);
/* autoregressive_latent_graph + compute + extract "cur" (main.cpp:5286-5352). codes: [B][502].
 * Only the first n_mel (<=502) mel positions are evaluated (causal => identical rows);
 * latents_out: [B][min(500,n_mel)][1024]. */
int tts_ar_latents(tts_ctx *ctx, const int32_t *codes502, int n_candidates, int n_mel,
                   float *latents_out);
/* process_logits_and_sample (main.cpp:4753-4806) on host logits with the ctx RNG:
 * penalty 2.0 on `penalty_ids` ([B][ids_per_cand]), temperature .8, top-k 50, top-p .8,
 * multinomial (2 draws). */
int tts_sample(tts_ctx *ctx, const float *logits, const int32_t *penalty_ids, int ids_per_cand,
               int n_candidates, int32_t *samples_out);
/* One decode step + its sampling in one call = tts_ar_step followed by tts_sample(penalty_ids = prev_ids, ids_per_cand = 1) (main.cpp:5227-5247 +
 * 4753-4806; flags & TTS_AR_MASK_STOP: logit 8193 forced to -1e30 first), with the sampler's top-k selected on the DEVICE: per candidate only the
 * 64..128 largest logits cross PCIe (16 KB per step of 16 candidates instead of 524 KB) and the host runs the same float tail over them — the ids
 * and the RNG consumption are those of the two-call sequence, bit for bit. A candidate whose list cannot decide (ties around the cut, see
 * host_logic.cpp: sample_one_list) is sampled from its full row, fetched on demand; tts_ar_topk_fallbacks counts those (candidates x steps of the last
 * tts_ar_step_sample / tts_autoregressive call). tts_autoregressive's loop runs on this path unless option "device_topk" is 0. */
/* After a tts_ar_step_sample call the host copy of the logits is UNDEFINED: only the rows the sampler had to fetch in full were refreshed (the pinned buffer is shared),
 * the others still hold an earlier step's values. A caller that needs the logits uses tts_ar_step. Fails with TTS_ERR_STATE before tts_ar_begin. */
int tts_ar_step_sample(tts_ctx *ctx, const int32_t *prev_ids, int step_i, unsigned flags, int32_t *samples_out);
int tts_ar_topk_fallbacks(const tts_ctx *ctx);
/* The whole autoregressive() driver (main.cpp:5042-5367): prefill, sample/decode loop with the
 * reference's stop rule, apply_padding, latent pass, trim_latents.
 *   flags: TTS_AR_MASK_STOP -> stop token never sampled, exactly max_steps codes (bench workload).
 *   codes_out [B][502]; rows_out [B] trimmed latent rows; latents_out: the trimmed latents of all
 *   candidates back to back (capacity B*500*1024 floats); steps_out: sampling iterations run. */
/*          TTS_AR_RETIRE (throughput mode) -> a candidate retires at its first 8193 and
 *          the loop ends when all have retired (the reference ends only when all B samples of ONE step are 8193,
 *          main.cpp:5214-5222); reaching max_steps pads the unfinished sequences and returns TTS_OK — which candidates
 *          were cut is reported by tts_ar_stop_status. No sequence differs from strict mode:
 *          sequences freeze at the first 8193 (5210-5213) and the uniforms are consumed identically. */
enum { TTS_AR_MASK_STOP = 1, TTS_AR_RETIRE = 2 };
int tts_autoregressive(tts_ctx *ctx, const int32_t *text_ids, int n_text, const float *voice1024,
                       int n_candidates, int max_steps, unsigned flags, int32_t *codes_out,
                       int32_t *rows_out, float *latents_out, int32_t *steps_out);
/* Stop schedule (benchmark / test device; random-init weights never sample a stop token, trained ones stop at different steps per candidate —
 * main.cpp:5188-5249): candidate b of the following tts_autoregressive calls samples the stop token 8193 at iteration stop_at[b] (= after stop_at[b]
 * codes) whatever its logits say; the uniforms are consumed as always. It applies ONLY to calls that pass TTS_AR_MASK_STOP | TTS_AR_RETIRE (round 6: any other
 * call ignores it and says so once on stderr — a forgotten schedule cannot truncate a strict run): the batch then becomes RAGGED in a reproducible way (decode steps
 * with retired candidates, a latent pass / diffusion row space / vocoder batch of unequal lengths). stop_at == NULL or n_candidates == 0 clears it; a call whose
 * candidate count differs from the schedule's fails with TTS_ERR_ARG before any device work. */
int tts_ar_set_stop_schedule(tts_ctx *ctx, const int32_t *stop_at, int n_candidates);
/* Per candidate of the last tts_autoregressive call: 1 = the sequence ends in a sampled stop token (what main.cpp:5214-5222
 * waits for), 0 = it was cut at max_steps (TTS_AR_RETIRE / TTS_AR_MASK_STOP) and padded like a finished one. */
int tts_ar_stop_status(tts_ctx *ctx, int32_t *stopped_out, int n_candidates);

/* ---- candidate re-ranking (not in the reference) -------------------------------------------- */
/* Score of every candidate = cosine similarity of the text latent and the candidate's speech-code latent x exp(temperature); the
 * caller keeps the arg-max (upstream tortoise-tts api.py; the reference keeps candidate 0, main.cpp:6575).
 * text_ids[n_text]: tokenizer output (ids < 256). codes: candidate c's sampled mel codes at codes[c * code_stride .. + code_len[c]),
 * every one < 8192 — the start token 8192 and the stop token 8193 are not scored (with the [B][502] output of tts_autoregressive:
 * codes + 1, code_stride = 502, code_len[c] = number of sampled codes before the stop token). scores_out[n_candidates]. */
int tts_clvp_score(tts_ctx *ctx, const int32_t *text_ids, int n_text, const int32_t *codes, const int32_t *code_len,
                   int n_candidates, int code_stride, float *scores_out);

/* ---- diffusion stage ----------------------------------------------------------------------- */
/* T = L*4*24000/22050 (main.cpp:5616-5617) */
int tts_diffusion_frames(int latent_rows);
/* One diffusion_graph evaluation (main.cpp:5749-5841 cond / 5866-5961 uncond): inputs
 * input_latent_tensor [L][1024], noise_tensor = x_t [100][T], timestep (raw 0..3999 value whose
 * sinusoidal embedding the reference uploads as time_embedding_{i}); conditioning_free as the
 * graph flag; out [200][T]. */
int tts_diffusion_forward(tts_ctx *ctx, const float *latents, int latent_rows, const float *x_t,
                          int timestep, int conditioning_free, float *out);
/* diffusion() for n_candidates independent latents (main.cpp:5614-6042; the reference runs one).
 *   latents: candidates back to back, rows[c] rows each; n_steps (reference: 80).
 *   noise: NULL -> noise_mode decides; else [sum_c (n_steps+1)*100*T_c] floats, per candidate
 *   x_T followed by one vector per step (the reference draws the last one too, 6020-6021).
 *   noise_mode (when noise==NULL): TTS_NOISE_REFERENCE draws from the ctx RNG in the reference's
 *   order, candidate by candidate; TTS_NOISE_DEVICE uses a counter-based device generator
 *   (seed = ctx seed, stream = candidate) — not the reference's noise, documented in DESIGN.md.
 *   mel_out: per candidate [100][T_c] back to back. */
enum { TTS_NOISE_REFERENCE = 0, TTS_NOISE_DEVICE = 1 };
int tts_diffusion(tts_ctx *ctx, const float *latents, const int32_t *rows, int n_candidates,
                  int n_steps, const float *noise, int noise_mode, float *mel_out);
/* The timestep MLP (main.cpp:3331-3343, 3410-3428: five tiny launches at the start of every tts_diffusion / tts_diffusion_forward call) is evaluated twice
 * and compared bit for bit, and repeated when the two evaluations disagree: a tripwire kept from round 4, when an earlier form of that kernel (packed f32 FMAs)
 * returned wrong sums while a second engine process used the same GPU. Number of disagreeing evaluations since the context was created:
 * 0 in every single-process run, and 0 beside a second process since the kernel was rebuilt. */
int tts_diffusion_time_mlp_retries(const tts_ctx *ctx);
/* Parity hardening (round 6): with option "fp16_check" = 1 every fp16 GEMM operand the diffusion stage produces (GroupNorm outputs, q | k rows, V^T columns,
 * attention outputs) is scanned after the launch that wrote it; split-precision weights are checked when they are packed (tts_load_diffusion).
 * counts[0] = non-finite values, counts[1] = values with |x| > 60000 (fp16 saturates at 65504) seen since the option was set. Returns TTS_OK.
 * The reference keeps these tensors in F32 (main.cpp:3191-3499, 3848-3875): an fp16 operand that saturates is a parity failure the tolerance tests on
 * small-sigma weights cannot see. */
int tts_diffusion_fp16_check(tts_ctx *ctx, int64_t counts[2]);

/* ---- vocoder stage -------------------------------------------------------------------------- */
int tts_vocoder_samples(int mel_frames); /* (T+10)*256-6, main.cpp:6051, 4459-4478 */
/* vocoder() (main.cpp:6044-6127): mel per candidate [100][T_c] (normalised, as returned by
 * tts_diffusion); noise NULL (see noise_mode above) or per candidate [64][T_c+10];
 * audio_out per candidate tts_vocoder_samples(T_c) floats back to back. */
int tts_vocoder(tts_ctx *ctx, const float *mel, const int32_t *frames, int n_candidates,
                const float *noise, int noise_mode, float *audio_out);

/* Streaming form for ONE candidate (first-audio latency; the reference has only the whole-utterance call): the samples of
 * frames [frame0, frame0 + n_frames) of the padded sequence (T + 10 frames; sample index = frame * 256, the sequence has (T+10)*256-6
 * samples). mel [100][T] and noise [64][T+10] are the WHOLE utterance's (draw the noise once with tts_rng_normal(ctx, buf, 64*(T+10)):
 * that is vocoder()'s draw, main.cpp:6058-6059); only a window with a fixed halo is evaluated. Concatenating the chunks of any
 * partition of [0, T+10) reproduces tts_vocoder's samples (same arithmetic per sample; tests/test_vocoder_gpu.py).
 * audio_out: capacity n_frames*256 floats; *n_samples_out: samples written. */
int tts_vocoder_chunk(tts_ctx *ctx, const float *mel, int mel_frames, const float *noise, int frame0, int n_frames,
                      float *audio_out, int *n_samples_out);

/* ---- output -------------------------------------------------------------------------------- */
/* writeWav (main.cpp:4821-4868): RIFF, fmt 16 B, tag 3 (IEEE float), mono, 32-bit. */
int tts_write_wav(const char *path, const float *samples, int64_t n, int sample_rate);

/* ---- host-logic probes ---------------------------------------------------------------------- */
/* The host-side arithmetic of the diffusion and AR drivers, callable without a device so that it can be checked
 * against the reference's own code on any machine. Not needed by an integration.
 * tts_host_schedule: respaced schedule + per-step scalars exactly as tts_diffusion uses them (main.cpp:5370-5493,
 *   5641-5716, 5988-6015); every output is [n_steps]. max_log = log(beta_t), min_log = posterior log variance (clipped),
 *   cfk = conditioning-free k, coef1/coef2 = posterior mean coefficients.
 * tts_host_timestep_embedding: main.cpp:5496-5521. tts_host_rel_bucket: main.cpp:4722-4749.
 * tts_host_pad_codes: apply_padding, main.cpp:4510-4532 (n <= 500 sampled codes -> 502). tts_host_trimmed_rows: trim_latents
 *   row count, main.cpp:4873-4915. */
int tts_host_schedule(int n_steps, int32_t *timestep_map, float *max_log, float *min_log, float *cfk, float *sqrt_recip,
                      float *sqrt_recipm1, float *coef1, float *coef2);
void tts_host_timestep_embedding(int t, float *out1024);
int tts_host_rel_bucket(int query, int key);
/* The sampler as a pure function of one candidate's row and its used uniform (what tts_sample runs per candidate), and the same decision taken
 * from a host restatement of the device prefilter's list (the `keep` <= 128 largest logits plus the ties of the smallest): the sampled id, or -1
 * where the engine would fetch the full row. For tests of the list logic without a GPU. */
int tts_host_sample_row(const float *row8194, const int32_t *penalty_ids, int n_ids, float uniform);
int tts_host_sample_prefiltered(const float *row8194, const int32_t *penalty_ids, int n_ids, float uniform, int keep);
int tts_host_pad_codes(const int32_t *codes, int n, int32_t *out502);
int tts_host_trimmed_rows(const int32_t *codes502);
/* Mel front-end of the two voice-conditioning encoders (host arithmetic, f64 inside; no counterpart in the reference, which has no audio input):
 * STFT n_fft = win = 1024, hop 256, periodic Hann, centre = true with reflect padding, frames = n / 256 + 1 (tts_host_mel_frames); n > 512.
 * tts_host_mel_diffusion100: 24 kHz audio -> [100][frames], upstream TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000) magnitude mel (librosa
 *   Slaney filterbank), log(clamp 1e-5). normalize = 0: as is = the input of tts_diffusion_conditioning_latent (upstream get_conditioning_latents:
 *   wav_to_univnet_mel(..., do_normalization = False)); normalize = 1: mapped to [-1, 1] (normalize_tacotron_mel) = the scale of the diffusion
 *   stage's output, which the vocoder driver de-normalises (main.cpp:6044-6060).
 * tts_host_mel_voice80: 22.05 kHz audio -> [80][frames], torchaudio MelSpectrogram(power 2, f_max 8000, norm "slaney", HTK mel scale),
 *   log(clamp 1e-5), divided per band by mel_norms80 (upstream's data/mel_norms.pth; NULL = no division) = the input of tts_voice_latent.
 * Return the frame count or a negative status. */
int tts_host_mel_frames(int64_t n_samples);
int tts_host_mel_diffusion100(const float *audio24k, int64_t n, int normalize, float *mel_out);
int tts_host_mel_voice80(const float *audio22k, int64_t n, const float *mel_norms80, float *mel_out);
/* The weight quantiser of option ar_weights = 2: OCP fp8 e4m3 (1-4-3, bias 7, largest finite 448, no infinities) code of v, round to
 * nearest even, saturating; NaN -> 0x7f. No counterpart in the reference. */
uint8_t tts_host_fp8_e4m3(float v);

/* ---- measurement hooks (bench.py; not part of the reference seam) -------------------------- */
/* Accumulated device time (ms; HIP event pairs recorded on the ctx stream around every launch, resolved
 * lazily so the timed region is not synchronised) and launch count of the named kernel family since the
 * last reset: "ar_gemv", "ar_attention", "ar_decode_step" (one whole decode-step graph replay; work = bytes streamed), "diff_gemm",
 * "diff_attn", "diff_gn_fused", "voc_lvc", ... The diffusion GEMM launches are recorded per shape class ("diff_gemm_qkv", "diff_gemm_k3", "diff_gemm_k3r",
 * "diff_gemm_k1", "diff_gemm_k1r", "diff_gemm_misc"); a family name also names its sub-families, so "diff_gemm" returns their sum
 * (tts_prof_get) and selects all of them ("prof_only:diff_gemm").
 * The totals cover the BRACKETED launches only: every "prof_stride"-th launch of a family, and — for the "diff_*" families, whose step
 * otherwise replays a captured hipGraph in which an event record would become a node — only the launches of the steps that run eagerly
 * (every "prof_eager_every"-th diffusion step while such a family is selected, 8 by default). ms / launches is therefore a per-launch
 * average over a sample; it is not the stage's total time (bench.py times stages with its own host clock around synchronised calls). */
int tts_prof_reset(tts_ctx *ctx, int enable);
/* work_out: summed algorithmic work of those launches — FLOPs for the MFMA-bound families (diff_gemm,
 * diff_attn, voc_kernel_gemm), bytes for the HBM-bound ones (ar_gemv: weight bytes streamed). */
int tts_prof_get(tts_ctx *ctx, const char *family, double *ms_out, int64_t *launches_out, double *work_out);

#ifdef __cplusplus
}
#endif
#endif /* TORTOISE_MI355X_H */
