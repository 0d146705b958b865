/* wesep_engine.h -- C ABI of libwesep_engine.so, the native (C++) inference runtime of wesep_amd for MI355X.
 *
 * MI355X counterpart of the reference's C++ runtime:
 *   runtime/separate/separate_engine.h:31-57  class SeparateEngine {ctor(model_path, feat_dim, sample_rate),
 *                                              ExtractFeature, ApplyMean, ForwardFunc(mix, spk1, spk2, output)}
 *   runtime/separate/separate_engine.cc:37-123 (TorchScript module on LibTorch-CPU, kaldi fbank + CMN on the host)
 * and of the whole-utterance forward of wesep/bin/infer.py:94-118.
 *
 * The reference loads a TorchScript archive; this engine loads a flat weight container written by
 * `python -m wesep_amd.bin.export_engine` (the `state_dict` under the reference's own key names, see INTEGRATION.md;
 * meta key "arch": 0 pBSRNN, 1 Conv-TasNet / SpEx+, 2 DPCCN, 3 TF-GridNet -- ws_engine_info(e, "arch")) and runs the forward as a
 * fixed launch plan over include/wesep_hip.h: weights are uploaded and
 * packed into MFMA fragment order ONCE at load, activations live in one grow-only device arena with stack
 * discipline, the enrollment front-end (kaldi fbank + CMN as two GEMMs, include/wesep_hip.h) and the jointly
 * trained ResNet speaker encoder (eval mode: BatchNorm folded to its running statistics) run on the device too.
 * Host buffers in, host buffers out; the engine owns its HIP stream and device memory.  One engine per GPU and
 * thread.  Return value: 0 or a negative WS_ERR_* code of wesep_hip.h, message via ws_engine_last_error().
 */
#ifndef WESEP_ENGINE_H_
#define WESEP_ENGINE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WS_ENGINE_ABI_VERSION 1

typedef struct ws_engine ws_engine;

/* flags of ws_engine_create */
#define WS_ENGINE_DRY_RUN 1 /* no GPU needed: host memory stands in for device memory and every launch is allowed to
                               fail with WS_ERR_LAUNCH, but every entry point's ARGUMENT validation must pass
                               (WS_ERR_INVALID still aborts).  Validates a weight container and the launch plan of a
                               given geometry on a machine without a GPU (refused when a HIP device is visible, where
                               the launches would execute); computes nothing (outputs are left untouched). */

int ws_engine_abi_version(void);
const char* ws_engine_last_error(void);

/* Loads the container, uploads and packs the weights on HIP device `device`.
 * Replaces SeparateEngine::SeparateEngine (separate_engine.cc:37-51: torch::jit::load + feature pipeline setup). */
int ws_engine_create(const char* weights_path, int device, int flags, ws_engine** out);
void ws_engine_destroy(ws_engine* e);

/* Model facts read from the container: key in {"sample_rate", "num_repeat", "spk_emb_dim", "joint_training",
 * "feat_dim", "n_tensors", "n_launches" (entry-point calls issued by the last forward), "arena_bytes",
 * "cluster_fallbacks" (forwards so far in which a weight-stationary cluster recurrence timed out -- its workgroups were
 * not co-resident, e.g. several engines on one GPU -- and the predicated streaming kernels recomputed the layer;
 * wesep_hip.h, ws_lstm_fwd_cluster)}; unknown key -> -1. */
long long ws_engine_info(const ws_engine* e, const char* key);

/* enrollment kinds */
#define WS_ENROLL_EMBEDDING 0 /* float [R][spk_emb_dim]: fixed speaker embeddings (joint_training = False models) */
#define WS_ENROLL_FBANK 1     /* float [R][enroll_len][feat_dim]: mean-normalised fbank (joint models, spk_feat True) */
#define WS_ENROLL_WAVE 2      /* float [R][enroll_len] in [-1, 1].  spk_feat = True models: kaldi fbank (dither 0) + CMN
                                 computed on the device (what SeparateEngine::ExtractFeature does on the host,
                                 separate_engine.cc:53-74); spk_feat = False models: their in-model PreEmphasis +
                                 MelSpectrogram + log + CMN front-end (bsrnn.py:343-350) */

/* est[r][0..T) = target-speaker estimate for mix[r][0..T) given enrollment r.  All pointers are HOST pointers.
 * Replaces `model(features, enroll)[0]` of infer.py:101-103 (whole utterance, any T >= 512; no chunking). */
int ws_engine_separate(ws_engine* e, const float* mix, int R, int T, const void* enroll, int enroll_kind,
                       int enroll_len, float* est);

/* The reference runtime's call: one mixture, two enrollment utterances (int16 PCM), two estimates.
 * mix [n] int16; spk1 / spk2 [n_enroll] int16; out [2][n] float in [-1, 1] like the reference (it scales the mixture by
 * 2^-15 before the model, separate_engine.cc:81-84, and its wav writer scales back, frontend/wav.h:245-253).
 * Replaces SeparateEngine::ForwardFunc (separate_engine.cc:76-123). */
int ws_engine_forward_pcm16(ws_engine* e, const int16_t* mix, int n, const int16_t* spk1, const int16_t* spk2,
                            int n_enroll, float* out);

#ifdef __cplusplus
}
#endif
#endif /* WESEP_ENGINE_H_ */
