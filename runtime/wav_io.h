// Minimal RIFF/WAVE PCM-16 reader and writer for the runtime's command-line tool.
// (The reference reads wavs with its own frontend/wav.h; this is an independent implementation of the same file
// format: canonical 44-byte headers plus tolerant chunk skipping on read.)
#ifndef WESEP_RUNTIME_WAV_IO_H_
#define WESEP_RUNTIME_WAV_IO_H_

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

namespace wesep_rt {

struct Wav {
  int sample_rate = 0;
  int channels = 0;
  std::vector<int16_t> samples;   // channel 0 only
};

inline bool read_wav(const std::string& path, Wav* out, std::string* err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    *err = "cannot open " + path;
    return false;
  }
  auto fail = [&](const char* why) {
    *err = path + ": " + why;
    fclose(f);
    return false;
  };
  char id[4];
  uint32_t size = 0;
  if (fread(id, 1, 4, f) != 4 || memcmp(id, "RIFF", 4) || fread(&size, 4, 1, f) != 1 || fread(id, 1, 4, f) != 4 ||
      memcmp(id, "WAVE", 4))
    return fail("not a RIFF/WAVE file");
  bool have_fmt = false;
  uint16_t format = 0, channels = 0, bits = 0;
  uint32_t rate = 0;
  while (fread(id, 1, 4, f) == 4 && fread(&size, 4, 1, f) == 1) {
    if (!memcmp(id, "fmt ", 4)) {
      uint8_t buf[16];
      if (size < 16 || fread(buf, 1, 16, f) != 16) return fail("short fmt chunk");
      memcpy(&format, buf, 2);
      memcpy(&channels, buf + 2, 2);
      memcpy(&rate, buf + 4, 4);
      memcpy(&bits, buf + 14, 2);
      if (size > 16) fseek(f, size - 16 + (size & 1), SEEK_CUR);
      have_fmt = true;
    } else if (!memcmp(id, "data", 4)) {
      if (!have_fmt) return fail("data chunk before fmt chunk");
      if (format != 1 || bits != 16 || channels < 1) return fail("only 16-bit PCM is supported");
      // the header's size is untrusted (streamed files carry 0xFFFFFFFF, crafted ones anything): never allocate more
      // than the file still holds
      size_t remaining = 0;
      {
        const long here = ftell(f);
        if (here >= 0 && fseek(f, 0, SEEK_END) == 0) {
          const long end = ftell(f);
          if (end > here) remaining = static_cast<size_t>(end - here);
          fseek(f, here, SEEK_SET);
        }
      }
      const size_t bytes = size < remaining ? size : remaining;
      const size_t frames = bytes / (2u * channels);
      std::vector<int16_t> raw(frames * channels);
      const size_t got = fread(raw.data(), 2, raw.size(), f) / channels;   // tolerate truncated files
      out->samples.resize(got);
      for (size_t i = 0; i < got; ++i) out->samples[i] = raw[i * channels];
      out->sample_rate = static_cast<int>(rate);
      out->channels = channels;
      fclose(f);
      return true;
    } else {
      fseek(f, size + (size & 1), SEEK_CUR);
    }
  }
  return fail("no data chunk");
}

// data in [-1, 1] -> 16-bit PCM, rounded and saturated
inline bool write_wav(const std::string& path, const float* data, size_t n, int sample_rate, std::string* err) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) {
    *err = "cannot create " + path;
    return false;
  }
  std::vector<int16_t> pcm(n);
  for (size_t i = 0; i < n; ++i) {
    float v = data[i] * 32768.0f;
    v = v > 32767.0f ? 32767.0f : (v < -32768.0f ? -32768.0f : v);
    pcm[i] = static_cast<int16_t>(v >= 0.f ? v + 0.5f : v - 0.5f);
  }
  const uint32_t bytes = static_cast<uint32_t>(n * 2), riff = 36 + bytes, fmt_size = 16, rate = sample_rate,
                 byte_rate = rate * 2;
  const uint16_t format = 1, channels = 1, align = 2, bits = 16;
  bool ok = fwrite("RIFF", 1, 4, f) == 4 && fwrite(&riff, 4, 1, f) == 1 && fwrite("WAVEfmt ", 1, 8, f) == 8 &&
            fwrite(&fmt_size, 4, 1, f) == 1 && fwrite(&format, 2, 1, f) == 1 && fwrite(&channels, 2, 1, f) == 1 &&
            fwrite(&rate, 4, 1, f) == 1 && fwrite(&byte_rate, 4, 1, f) == 1 && fwrite(&align, 2, 1, f) == 1 &&
            fwrite(&bits, 2, 1, f) == 1 && fwrite("data", 1, 4, f) == 4 && fwrite(&bytes, 4, 1, f) == 1 &&
            fwrite(pcm.data(), 2, n, f) == n;
  fclose(f);
  if (!ok) *err = "short write to " + path;
  return ok;
}

}  // namespace wesep_rt
#endif  // WESEP_RUNTIME_WAV_IO_H_
