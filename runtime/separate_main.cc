// separate_main -- command-line front end of the native runtime, with the interface of the reference's
// runtime/bin/separate_main.cc:24-115:
//
//   separate_main --wav_scp scp --model model.wsw --output_dir out [--sample_rate 16000] [--devices 0,1] [--jobs 4]
//                 [--dry_run]
//   separate_main --wav_path mix.wav --spk1_emb e1.wav --spk2_emb e2.wav --model model.wsw --output_dir out
//
// wav_scp lines: "<key> <mixture.wav> <enroll_spk1.wav> <enroll_spk2.wav>".  For every line the mixture and the two
// enrollment utterances go through ws_engine_forward_pcm16 (enrollments cut to the shorter one, as the reference does)
// and <key>-spk1.wav / <key>-spk2.wav are written; the real-time factor is printed per utterance and in total.
// Utterances are independent: --jobs J worker threads, each with its own engine (own HIP stream, arena and weight
// copy -- 0.3 GB of 288), spread round-robin over --devices.  Engines that share a GPU overlap on the device (round 2
// serialised them; round 3 removed the cause -- profiles/r03_kernel_race.md -- and WS_ENGINE_SERIALIZE=1 restores one
// forward at a time per GPU).  The reference tool is single-threaded on CPU cores.
// --dry_run validates the model file and the launch plan of every utterance without a GPU and writes nothing.
// --raw_out additionally writes the unquantised estimates as <key>-spk{1,2}.f32 (float32, for parity checks).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/wesep_engine.h"
#include "wav_io.h"

namespace {

struct Args {
  std::map<std::string, std::string> kv;
  const std::string& get(const std::string& k, const std::string& dflt) const {
    auto it = kv.find(k);
    return it == kv.end() ? dflt : it->second;
  }
  bool has(const std::string& k) const { return kv.count(k) != 0; }
};

// gflags-style: --name=value, --name value, --flag
Args parse(int argc, char** argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    if (s.rfind("--", 0) != 0) continue;
    s = s.substr(2);
    const size_t eq = s.find('=');
    if (eq != std::string::npos) {
      a.kv[s.substr(0, eq)] = s.substr(eq + 1);
    } else if (i + 1 < argc && strncmp(argv[i + 1], "--", 2) != 0) {
      a.kv[s] = argv[++i];
    } else {
      a.kv[s] = "true";
    }
  }
  return a;
}

int die(const std::string& msg) {
  fprintf(stderr, "separate_main: %s\n", msg.c_str());
  return 1;
}

}  // namespace

int main(int argc, char** argv) {
  const Args args = parse(argc, argv);
  const std::string model = args.get("model", ""), out_dir = args.get("output_dir", "");
  const int sample_rate = atoi(args.get("sample_rate", "16000").c_str());
  const bool dry = args.has("dry_run"), raw_out = args.has("raw_out");
  if (model.empty()) return die("--model is required");
  if (out_dir.empty() && !dry) return die("Invalid output path.");

  std::vector<std::vector<std::string>> waves;
  if (args.has("wav_path") && args.has("spk1_emb") && args.has("spk2_emb")) {
    waves.push_back({"test", args.get("wav_path", ""), args.get("spk1_emb", ""), args.get("spk2_emb", "")});
  } else {
    std::ifstream scp(args.get("wav_scp", ""));
    std::string line;
    while (std::getline(scp, line)) {
      std::istringstream is(line);
      std::vector<std::string> f;
      std::string tok;
      while (is >> tok) f.push_back(tok);
      if (f.empty()) continue;
      if (f.size() != 4) return die("wav_scp line needs 4 fields: key mix spk1 spk2 -- got: " + line);
      waves.push_back(f);
    }
    if (waves.empty()) return die("Please provide non-empty wav scp.");
  }

  std::vector<int> devices;
  {
    std::istringstream ds(args.get("devices", args.get("device", "0")));
    std::string tok;
    while (std::getline(ds, tok, ',')) devices.push_back(atoi(tok.c_str()));
    if (devices.empty()) devices.push_back(0);
  }
  int jobs = atoi(args.get("jobs", "1").c_str());
  if (jobs < 1) jobs = 1;
  if (jobs > static_cast<int>(waves.size())) jobs = static_cast<int>(waves.size());

  std::mutex io_mu;                 // stdout / first error
  std::string first_error;
  std::atomic<size_t> next{0};
  std::atomic<bool> failed{false};
  double total_audio_ms = 0.0, total_busy_ms = 0.0;
  auto fail = [&](const std::string& msg) {
    std::lock_guard<std::mutex> l(io_mu);
    if (first_error.empty()) first_error = msg;
    failed = true;
  };
  auto worker = [&](int job) {
    ws_engine* engine = nullptr;
    long long fallbacks_seen = 0;
    if (ws_engine_create(model.c_str(), devices[job % devices.size()], dry ? WS_ENGINE_DRY_RUN : 0, &engine) != 0)
      return fail(ws_engine_last_error());
    if (ws_engine_info(engine, "sample_rate") != sample_rate) {
      ws_engine_destroy(engine);
      return fail("model sample rate differs from --sample_rate");
    }
    for (size_t i = next++; i < waves.size() && !failed; i = next++) {
      const auto& w = waves[i];
      wesep_rt::Wav mix, s1, s2;
      std::string err;
      if (!wesep_rt::read_wav(w[1], &mix, &err) || !wesep_rt::read_wav(w[2], &s1, &err) ||
          !wesep_rt::read_wav(w[3], &s2, &err)) {
        fail(err);
        break;
      }
      if (mix.sample_rate != sample_rate || s1.sample_rate != sample_rate || s2.sample_rate != sample_rate) {
        fail(w[0] + ": sample rate is not " + std::to_string(sample_rate));
        break;
      }
      const int n = static_cast<int>(mix.samples.size());
      const int n_enroll = static_cast<int>(s1.samples.size() < s2.samples.size() ? s1.samples.size() : s2.samples.size());
      std::vector<float> out(size_t(2) * n, 0.f);
      const auto t0 = std::chrono::steady_clock::now();
      if (ws_engine_forward_pcm16(engine, mix.samples.data(), n, s1.samples.data(), s2.samples.data(), n_enroll,
                                  out.data()) != 0) {
        fail(w[0] + ": " + ws_engine_last_error());
        break;
      }
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      const double audio_ms = 1000.0 * n / sample_rate;
      if (!dry && (!wesep_rt::write_wav(out_dir + "/" + w[0] + "-spk1.wav", out.data(), n, sample_rate, &err) ||
                   !wesep_rt::write_wav(out_dir + "/" + w[0] + "-spk2.wav", out.data() + n, n, sample_rate, &err))) {
        fail(err);
        break;
      }
      if (!dry && raw_out) {
        for (int k = 0; k < 2; ++k) {
          FILE* f = fopen((out_dir + "/" + w[0] + "-spk" + std::to_string(k + 1) + ".f32").c_str(), "wb");
          if (!f || fwrite(out.data() + size_t(k) * n, 4, n, f) != size_t(n)) fail("cannot write raw output");
          if (f) fclose(f);
        }
      }
      std::lock_guard<std::mutex> l(io_mu);
      printf("process: %s RTF: %.4f (%lld launches, %lld MiB arena)%s\n", w[0].c_str(), ms / audio_ms,
             ws_engine_info(engine, "n_launches"), ws_engine_info(engine, "arena_bytes") >> 20, dry ? " [dry run]" : "");
      if (ws_engine_info(engine, "cluster_fallbacks") > fallbacks_seen) {
        fallbacks_seen = ws_engine_info(engine, "cluster_fallbacks");
        printf("note: %s: a cluster recurrence timed out (GPU shared); recomputed by the streaming kernels\n", w[0].c_str());
      }
      total_audio_ms += audio_ms;
      total_busy_ms += ms;
    }
    ws_engine_destroy(engine);
  };
  const auto wall0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int j = 1; j < jobs; ++j) pool.emplace_back(worker, j);
  worker(0);
  for (auto& t : pool) t.join();
  if (failed) return die(first_error);
  const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  // with one job this is the reference tool's total (engine time only); with more, wall time is what counts
  const double taken = jobs == 1 ? total_busy_ms : wall_ms;
  printf("Total: process %.0fms audio taken %.0fms.\nRTF: %.4f\n", total_audio_ms, taken,
         total_audio_ms > 0 ? taken / total_audio_ms : 0.0);
  return 0;
}
