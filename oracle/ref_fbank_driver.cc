// TEST INFRASTRUCTURE -- oracle/_ref builder input, never linked into the product.
// C entry point around the reference's own kaldi-style filterbank front-end, compiled from the sources where they
// lie under /root/reference (runtime/frontend/fbank.h:31-222 `wenet::Fbank`, runtime/frontend/fft.cc), plus the
// per-utterance mean normalisation of runtime/separate/separate_engine.cc:63-74 (`ApplyMean`), restated below in
// four lines because that file needs libtorch.  Used to pin oracle/fbank_oracle.py and to generate
// tests/golden/fbank_*.npz (oracle/make_golden.py).
#include <vector>

#include "frontend/fbank.h"

extern "C" int ref_fbank(const float* wav, int num_samples, int num_bins, int sample_rate, int frame_length,
                         int frame_shift, int apply_mean, float* out /* [frames][num_bins] */, int max_frames) {
  wenet::Fbank fb(num_bins, sample_rate, frame_length, frame_shift);   // dither 0, remove_dc_offset, log
  std::vector<float> w(wav, wav + num_samples);
  std::vector<std::vector<float>> feat;
  int n = fb.Compute(w, &feat);
  if (n > max_frames) return -1;
  if (apply_mean && n > 0) {
    for (int j = 0; j < num_bins; ++j) {
      float m = 0.f;
      for (int i = 0; i < n; ++i) m += feat[i][j];
      m /= n;
      for (int i = 0; i < n; ++i) feat[i][j] -= m;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < num_bins; ++j) out[i * num_bins + j] = feat[i][j];
  return n;
}
