// `tortoise` command line — the reference's CLI surface on top of the C ABI.
//
// Same flags, defaults, relative paths and exit codes as main() in /root/reference/main.cpp:6528-6584:
//   --message <str>   default "this is a test message."
//   --voice <path>    default ../models/mol.bin        (1024 raw f32 auto_conditioning)
//   --output <path>   default ./output.wav             (24 kHz mono IEEE-float WAV)
//   --seed <int>      std::stoi -> mt19937::seed; absent => wall-clock ms
// Flags are matched as adjacent pairs at argv[i], i < argc-1; unknown flags are ignored.
// Weights are read from ../models/ggml-model.bin, ggml-diffusion-model.bin, ggml-vocoder-model.bin
// and the tokenizer from ../models/tokenizer.json (main.cpp:5078, 5625, 6046, 6551).
// Extensions (not in the reference): --models <dir>, --candidates <n> (all candidates are carried
// through; candidate 0 is written to --output like the reference, others to <output>.<c>.wav),
// --steps <n> diffusion steps (default 80), --device <ordinal>, --codes <n>,
// --clvp <file>: re-rank the candidates with CLVP (not in the reference, which keeps candidate 0, main.cpp:6575; upstream tortoise-tts
//   does this): every candidate's codes are scored against the text, only the best one goes through diffusion + vocoder and is written
//   to --output. With --devices every worker scores its own shard and the parent keeps the best of the workers' winners.
// --diffusion-latent <file>: 2048 raw f32 that replace the `diffusion_conditioning_latent` weight of ggml-diffusion-model.bin (the reference
//   bakes ONE voice into that file, main.cpp:1557-1560); with --voice this makes a voice two small files (tools/make_voice.py writes both).
// --devices <N> [--device-map a,b,...]: candidate-parallel multi-GPU run. The process re-executes itself once per GPU
//   (one process per device, replicated weights); worker r takes candidates [r B/N, (r+1) B/N) of the ONE batch: the RNG stream
//   partition (options rng_shard_offset / rng_shard_total) makes the N x B/N codes identical to a single-GPU run of B candidates,
//   device noise is keyed by the global candidate id, and the throughput stop rule (TTS_AR_RETIRE) needs no per-step exchange.
//   --exchange files (default): nothing is exchanged between workers — candidates never interact — each writes its own candidates' WAV
//   files (<output> for candidate 0, <output>.<c>.wav for the others, c = global candidate index).
//   --exchange rccl: the workers form one RCCL communicator (cli_rccl.h; distinct GPUs per worker): rank 0 broadcasts the conditioning
//   (text ids + voice latent), the result sizes / CLVP scores are all-gathered, the audio is sent to rank 0, which writes every WAV file.
#include "tortoise_mi355x.h"
#include "cli_rccl.h"
#include <algorithm>
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <thread>

static int die(tts_ctx *c, const char *what) {
  fprintf(stderr, "%s: %s\n", what, tts_last_error(c));
  return 1; // the reference exit(1)s on load failure (main.cpp:5089, 5634, 6069)
}

int main(int argc, char **argv) {
  std::string message = "this is a test message.";
  std::string voicePath = "../models/mol.bin";
  std::string outputPath = "./output.wav";
  std::string modelsDir = "../models";
  bool have_seed = false;
  int seed = 0, candidates = 1, steps = 80, device = 0, fixed_codes = 0, devices = 1, shard = -1, nshards = 1;
  std::string device_map, clvpPath, exchange = "files", rccl_id, diffLatentPath;
  bool dry = false, allow_shared = false;
  bool timing = false;
  int test_fail_shard = -1, test_slow_shard = -1;
  std::vector<std::pair<std::string, double>> engine_options; // --option key=value (repeatable): tts_set_option before the models are loaded
  for (int i = 1; i < argc - 1; ++i) {
    std::string a(argv[i]);
    if (a == "--voice") voicePath = argv[i + 1];
    else if (a == "--message") message = argv[i + 1];
    else if (a == "--output") outputPath = argv[i + 1];
    else if (a == "--seed") { seed = std::stoi(argv[i + 1]); have_seed = true; }
    else if (a == "--models") modelsDir = argv[i + 1];
    else if (a == "--candidates") candidates = std::stoi(argv[i + 1]);
    else if (a == "--steps") steps = std::stoi(argv[i + 1]);
    else if (a == "--device") device = std::stoi(argv[i + 1]);
    else if (a == "--codes") fixed_codes = std::stoi(argv[i + 1]); // exactly N sampled codes, stop token masked (synthetic weights never stop)
    else if (a == "--devices") devices = std::stoi(argv[i + 1]);
    else if (a == "--device-map") device_map = argv[i + 1];
    else if (a == "--allow-shared-device") allow_shared = argv[i + 1][0] != '0';
    else if (a == "--clvp") clvpPath = argv[i + 1];
    else if (a == "--exchange") exchange = argv[i + 1];
    else if (a == "--dry-run") dry = argv[i + 1][0] != '0'; // plumbing check without a device, see below
    else if (a == "--timing") timing = argv[i + 1][0] != '0'; // wall clock of every phase of this process on stderr
    else if (a == "--test-fail-shard") test_fail_shard = std::stoi(argv[i + 1]); // test hooks (tests/test_distributed_cpu.py): worker r exits with status 3 after the
    else if (a == "--test-slow-shard") test_slow_shard = std::stoi(argv[i + 1]); // conditioning broadcast / sleeps 2 s before the final exchange
    else if (a == "--option") { // engine option, e.g. --option attn_f32=1 (include/tortoise_mi355x.h: tts_set_option)
      const std::string kv = argv[i + 1];
      const size_t eq = kv.find('=');
      if (eq == std::string::npos) { fprintf(stderr, "--option %s: expected key=value\n", kv.c_str()); return 1; }
      engine_options.emplace_back(kv.substr(0, eq), std::atof(kv.c_str() + eq + 1));
    }
    else if (a == "--diffusion-latent") diffLatentPath = argv[i + 1];
    else if (a == "--rccl-id") rccl_id = argv[i + 1]; // worker mode (set by the parent)
    else if (a == "--shard") { // worker mode (set by the parent): "r/N"
      std::string v(argv[i + 1]);
      const size_t sl = v.find('/');
      if (sl != std::string::npos) { shard = std::stoi(v.substr(0, sl)); nshards = std::stoi(v.substr(sl + 1)); }
    }
  }
  if (exchange != "files" && exchange != "rccl") { fprintf(stderr, "--exchange %s: files or rccl\n", exchange.c_str()); return 1; }
  if ((devices > 1 || exchange == "rccl") && shard < 0) { // parent: one worker process per GPU
    if (candidates % devices) { fprintf(stderr, "--candidates %d does not divide over --devices %d\n", candidates, devices); return 1; }
    std::vector<int> map;
    for (size_t p = 0; p < device_map.size();) {
      const size_t q = device_map.find(',', p);
      map.push_back(std::stoi(device_map.substr(p, q == std::string::npos ? std::string::npos : q - p)));
      if (q == std::string::npos) break;
      p = q + 1;
    }
    // One engine process per GPU. Two on ONE device is not a deployment form: round 4 saw a kernel's packed f32 FMAs return wrong sums while another
    // process's MFMA waves shared the GPU (profiles/r4_two_process_determinism.txt). The default build KEEPS the packed f32 forms (a build without them is one make
    // variable away: `make PK=...`, profiles/r5_packed_f32_ab.txt); what protects a run is this refusal — and, with --exchange files, running the workers that share a
    // device one after the other. --allow-shared-device 1 with --exchange rccl still puts two engine processes on one GPU at once (tests on a one-GPU box only).
    bool shared = false;
    for (int r = 0; r < devices; r++)
      for (int q = 0; q < r; q++)
        if ((r < (int)map.size() ? map[r] : r) == (q < (int)map.size() ? map[q] : q)) {
          shared = true;
          if (!dry && !allow_shared) {
            fprintf(stderr, "--device-map %s: workers %d and %d would share a GPU (unsupported; --allow-shared-device 1 to run anyway)\n", device_map.c_str(), q, r);
            return 1;
          }
        }
    // --allow-shared-device with --exchange files: the workers run ONE AFTER THE OTHER, so no two engine processes are ever busy on the same GPU (an RCCL exchange
    // needs all ranks alive at once — and a distinct GPU per rank, which RCCL itself enforces)
    const bool serial = shared && !dry && exchange == "files";
    if (!have_seed) // every worker must draw from the same stream
      seed = (int)(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count() & 0x7fffffff);
    std::string id_hex;
    if (exchange == "rccl") { // the communicator's id is created here and handed to every worker
      if (dry && !getenv("TTS_RCCL_LIB")) {
        fprintf(stderr, "--dry-run with --exchange rccl needs TTS_RCCL_LIB=<stand-in library> (tests/fake_rccl.cpp): librccl cannot take the host buffers of a dry run\n");
        return 1;
      }
      RcclApi api;
      ncclUniqueId id;
      if (!api.open()) return 1;
      const ncclResult_t rr = api.GetUniqueId(&id);
      if (rr != ncclSuccess) { fprintf(stderr, "ncclGetUniqueId: %s\n", api.GetErrorString(rr)); return 1; }
      id_hex = rccl_id_to_hex(id);
    }
    std::vector<pid_t> pids;
    for (int r = 0; r < devices; r++) {
      const pid_t pid = fork();
      if (pid < 0) { perror("fork"); return 1; }
      if (pid == 0) {
        std::vector<std::string> args(argv, argv + argc);
        const std::string extra[] = {"--shard", std::to_string(r) + "/" + std::to_string(devices), "--device",
                                     std::to_string(r < (int)map.size() ? map[r] : r), "--seed", std::to_string(seed), "--end", "-"};
        if (!id_hex.empty()) { args.push_back("--rccl-id"); args.push_back(id_hex); }
        args.insert(args.end(), std::begin(extra), std::end(extra)); // later flags override earlier ones; "--end -" keeps the last pair inside i < argc-1
        std::vector<char *> av;
        for (auto &x : args) av.push_back(&x[0]);
        av.push_back(nullptr);
        execv("/proc/self/exe", av.data());
        perror("execv");
        _exit(127);
      }
      if (serial) {
        int st = 0;
        if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) return 1;
      } else pids.push_back(pid);
    }
    int rc = 0;
    std::vector<pid_t> alive = pids;
    while (!alive.empty()) {
      int st = 0;
      const pid_t pid = wait(&st);
      if (pid < 0) { rc = 1; break; }
      alive.erase(std::remove(alive.begin(), alive.end(), pid), alive.end());
      if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
        if (rc == 0 && exchange == "rccl")  // the workers still running would wait in a collective for ever
          for (pid_t p : alive) kill(p, SIGTERM);
        rc = 1;
      }
    }
    if (rc == 0 && !clvpPath.empty() && exchange == "files") { // every worker left "<output>.shard<r>.score" = "<global candidate> <score>" and that candidate's WAV
      int best_gc = -1;
      double best_score = 0;
      std::vector<int> winners;
      for (int r = 0; r < devices; r++) {
        const std::string sp = outputPath + ".shard" + std::to_string(r) + ".score";
        std::ifstream f(sp);
        int gc; double sc;
        if (!(f >> gc >> sc)) { fprintf(stderr, "missing CLVP score of worker %d\n", r); return 1; }
        winners.push_back(gc);
        if (best_gc < 0 || sc > best_score) { best_gc = gc; best_score = sc; }
        std::remove(sp.c_str());
      }
      for (int gc : winners) {
        const std::string wp = outputPath + "." + std::to_string(gc) + ".wav";
        if (gc == best_gc) { if (std::rename(wp.c_str(), outputPath.c_str())) { perror("rename"); return 1; } }
        else std::remove(wp.c_str());
      }
      printf("clvp: candidate %d kept (score %.5f)\n", best_gc, best_score);
      std::cout << "WAV file saved successfully. :^)" << std::endl;
    }
    return rc;
  }
  const int total_candidates = candidates;
  if (shard >= 0) candidates = total_candidates / nshards;
  using clk = std::chrono::steady_clock;
  const clk::time_point t_start = clk::now();
  clk::time_point t_last = t_start;
  tts_ctx *ctx = tts_create(dry ? -1 : device); // --dry-run: a host-only context (tokenizer, RNG, sampler; every stage call would fail)
  auto mark = [&](const char *what) { // --timing 1
    if (!timing) return;
    const clk::time_point t = clk::now();
    fprintf(stderr, "[timing] %-22s %8.1f ms (at %8.1f)\n", what, std::chrono::duration<double, std::milli>(t - t_last).count(),
            std::chrono::duration<double, std::milli>(t - t_start).count());
    t_last = t;
  };
  mark("tts_create");
  if (!ctx) {
    fprintf(stderr, "tts_create(%d) failed: no HIP device (this engine has no CPU path)\n", device);
    return 1;
  }
  if (have_seed) tts_seed(ctx, (uint32_t)seed);
  for (const auto &kv : engine_options)
    if (tts_set_option(ctx, kv.first.c_str(), kv.second)) return die(ctx, "--option");
  if (shard >= 0) {
    tts_set_option(ctx, "rng_shard_offset", (double)(shard * candidates));
    tts_set_option(ctx, "rng_shard_total", (double)total_candidates);
  }
  if (tts_tokenizer_load(ctx, (modelsDir + "/tokenizer.json").c_str()) < 0) return die(ctx, "tokenizer");
  std::vector<int32_t> tokens(4096);
  int n = tts_tokenize(ctx, message.c_str(), tokens.data(), (int)tokens.size());
  if (n < 0) return die(ctx, "tokenize");
  tokens.resize(n);

  std::vector<float> voice(1024);
  {
    std::ifstream f(voicePath, std::ios::binary);
    if (!f) { std::cerr << "Error: Unable to open file " << voicePath << std::endl; return 1; }
    f.read((char *)voice.data(), 1024 * sizeof(float));
  }
  RcclWorld world;
  const bool use_rccl = shard >= 0 && !rccl_id.empty();
  if (use_rccl) {
    if (!world.init(rccl_id, shard, nshards, dry)) { fprintf(stderr, "rccl: %s\n", world.err.c_str()); return 1; }
    // conditioning from rank 0: number of text ids, the ids, the voice latent (what SURVEY 8e's ncclBroadcast carries)
    struct { int32_t n; int32_t ids[4096]; float voice[1024]; } cond;
    memset(&cond, 0, sizeof cond);
    if (shard == 0) { cond.n = n; memcpy(cond.ids, tokens.data(), (size_t)n * 4); memcpy(cond.voice, voice.data(), 4096); }
    if (!world.broadcast(&cond, sizeof cond, 0)) { fprintf(stderr, "rccl: %s\n", world.err.c_str()); return 1; }
    n = cond.n;
    tokens.assign(cond.ids, cond.ids + n);
    memcpy(voice.data(), cond.voice, 4096);
    if (shard == test_fail_shard) { fprintf(stderr, "worker %d: --test-fail-shard\n", shard); _exit(3); } // the others are left waiting in the final exchange
  }
  if (shard >= 0 && shard == test_slow_shard) sleep(2);
  // results of the three stages (or of their stand-in under --dry-run): B output candidates, candidate c has nsamp[c] samples in `audio`
  int B = candidates, kept_gc = -1;
  double kept_score = 0;
  std::vector<float> audio;
  std::vector<size_t> nsamp;
  std::vector<int32_t> frames;
  const int B_ar = candidates;
  if (dry) {
    // Plumbing check without a device (tests/test_distributed_cpu.py): the host sampler on fixed synthetic logits stands in for the AR stage (so the
    // RNG stream partition of a --devices run is the real one), a candidate's "audio" is its sampled ids, its CLVP score a function of them. Exercises
    // the parent's fork / exec / --shard / --seed hand-over, the per-candidate file names, the parent's pick among the workers' winners and its exit code.
    const int nsteps_dry = fixed_codes > 0 ? fixed_codes : 8, V = TTS_VOCAB_MEL;
    std::vector<float> logits((size_t)B_ar * V);
    std::vector<int32_t> prev(B_ar, 8192), ids(B_ar);
    std::vector<std::vector<float>> seqs(B_ar);
    for (int st = 0; st < nsteps_dry; st++) {
      for (int c = 0; c < B_ar; c++)
        for (int v = 0; v < V; v++) logits[(size_t)c * V + v] = 3.0f * std::sin(0.37f * (float)v + 0.11f * (float)st); // the same for every candidate: only the draws differ
      if (tts_sample(ctx, logits.data(), prev.data(), 1, B_ar, ids.data())) return die(ctx, "sample");
      for (int c = 0; c < B_ar; c++) { seqs[c].push_back((float)ids[c]); prev[c] = ids[c]; }
    }
    int best = 0;
    std::vector<double> scores(B_ar, 0.0);
    for (int c = 0; c < B_ar; c++)
      for (float v : seqs[c]) scores[c] = std::fmod(scores[c] * 31.0 + (double)v, 1009.0);
    for (int c = 1; c < B_ar; c++)
      if (scores[c] > scores[best]) best = c;
    if (!clvpPath.empty()) {
      kept_gc = (shard >= 0 ? shard * B_ar : 0) + best; kept_score = scores[best]; B = 1;
      audio = seqs[best]; nsamp.assign(1, audio.size());
      if (shard < 0) printf("clvp: candidate %d kept (score %.5f)\n", kept_gc, kept_score);
    } else {
      for (int c = 0; c < B_ar; c++) { audio.insert(audio.end(), seqs[c].begin(), seqs[c].end()); nsamp.push_back(seqs[c].size()); }
    }
  } else {
  mark("tokenizer, voice");
  // The diffusion and vocoder models do not depend on anything the autoregressive stage produces: they are read, re-laid out and uploaded on a second thread while this
  // one loads and runs the autoregressive model (round 6; the reference loads each model in front of its stage, main.cpp:5089, 5634, 6069). Their status is looked at where
  // the reference would have loaded them, so a bad file is reported at the same point of the run with the same message.
  int rc_diff = 0, rc_voc = 0;
  std::string err_diff, err_voc;
  struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } bg;
  bg.t = std::thread([&]() {
    rc_diff = tts_load_diffusion(ctx, (modelsDir + "/ggml-diffusion-model.bin").c_str());
    if (rc_diff) { err_diff = tts_last_error(ctx); return; }
    rc_voc = tts_load_vocoder(ctx, (modelsDir + "/ggml-vocoder-model.bin").c_str());
    if (rc_voc) err_voc = tts_last_error(ctx);
  });
  if (tts_load_ar(ctx, (modelsDir + "/ggml-model.bin").c_str())) return die(ctx, "autoregressive_model_load");
  mark("load autoregressive");
  std::vector<int32_t> codes((size_t)B_ar * 502), rows(B_ar);
  std::vector<float> latents((size_t)B_ar * 500 * 1024);
  int32_t nsteps = 0;
  // More than one candidate (in this process or across --devices shards): the throughput stop rule. The reference's "all B samples of ONE
  // step are 8193" practically never fires for B > 1 (and would need a per-step exchange between shards); every sequence is the same.
  const unsigned ar_flags = (fixed_codes > 0 ? TTS_AR_MASK_STOP : 0) | (total_candidates > 1 ? TTS_AR_RETIRE : 0);
  if (tts_autoregressive(ctx, tokens.data(), n, voice.data(), B_ar, fixed_codes > 0 ? fixed_codes : 500, ar_flags,
                         codes.data(), rows.data(), latents.data(), &nsteps))
    return die(ctx, "autoregressive");
  mark("autoregressive");
  printf("tokens sampled: %d\n", nsteps);
  if (fixed_codes <= 0) {
    std::vector<int32_t> stopped(B_ar);
    if (tts_ar_stop_status(ctx, stopped.data(), B_ar) == 0)
      for (int c = 0; c < B_ar; c++)
        if (!stopped[c]) fprintf(stderr, "warning: candidate %d sampled no stop token within 500 codes (sequence cut)\n", (shard >= 0 ? shard * B_ar : 0) + c);
  }

  // CLVP re-ranking (extension): keep the candidate whose codes (the rows the diffusion stage would consume) score best against the text
  B = B_ar;
  const float *lat_in = latents.data();
  if (!clvpPath.empty()) {
    if (tts_load_clvp(ctx, clvpPath.c_str())) return die(ctx, "clvp_model_load");
    std::vector<float> scores(B_ar);
    if (tts_clvp_score(ctx, tokens.data(), n, codes.data() + 1, rows.data(), B_ar, 502, scores.data())) return die(ctx, "clvp");
    int best = 0;
    for (int c = 1; c < B_ar; c++)
      if (scores[c] > scores[best]) best = c;
    printf("clvp scores:");
    for (int c = 0; c < B_ar; c++) printf(" %.5f", scores[c]);
    printf("\n");
    size_t off_rows = 0;
    for (int c = 0; c < best; c++) off_rows += (size_t)rows[c];
    lat_in = latents.data() + off_rows * 1024;
    rows[0] = rows[best];
    B = 1;
    kept_gc = (shard >= 0 ? shard * B_ar : 0) + best;
    kept_score = scores[best];
    if (total_candidates > 1) { // device noise stays keyed by the kept candidate's global id
      tts_set_option(ctx, "rng_shard_offset", (double)kept_gc);
      tts_set_option(ctx, "rng_shard_total", (double)total_candidates);
    }
    if (shard < 0) printf("clvp: candidate %d kept (score %.5f)\n", kept_gc, kept_score);
  }

  bg.t.join();
  if (rc_diff) { fprintf(stderr, "diffusion_model_load: %s\n", err_diff.c_str()); return 1; }
  mark("wait for the diffusion + vocoder loads");
  if (!diffLatentPath.empty()) {
    std::vector<float> dl(2048);
    std::ifstream f(diffLatentPath, std::ios::binary);
    if (!f || !f.read((char *)dl.data(), 2048 * sizeof(float))) { std::cerr << "Error: Unable to read 2048 floats from " << diffLatentPath << std::endl; return 1; }
    if (tts_set_diffusion_conditioning_latent(ctx, dl.data())) return die(ctx, "diffusion_conditioning_latent");
  }
  size_t mel_total = 0, audio_total = 0;
  frames.assign(B, 0);
  for (int c = 0; c < B; c++) {
    frames[c] = tts_diffusion_frames(rows[c]);
    mel_total += (size_t)100 * frames[c];
    audio_total += (size_t)tts_vocoder_samples(frames[c]);
  }
  std::vector<float> mel(mel_total);
  audio.assign(audio_total, 0.f);
  // B == 1: the reference's exact RNG order (AR uniforms, x_T, per-step noise, vocoder noise)
  const int noise_mode = (total_candidates == 1) ? TTS_NOISE_REFERENCE : TTS_NOISE_DEVICE;
  if (tts_diffusion(ctx, lat_in, rows.data(), B, steps, nullptr, noise_mode, mel.data())) return die(ctx, "diffusion");
  mark("diffusion");
  if (tts_diffusion_time_mlp_retries(ctx) > 0) // only ever seen while another process shares the GPU (include/tortoise_mi355x.h)
    fprintf(stderr, "[tortoise] the timestep MLP was re-evaluated %d times before two evaluations agreed\n", tts_diffusion_time_mlp_retries(ctx));
  if (rc_voc) { fprintf(stderr, "vocoder_model_load: %s\n", err_voc.c_str()); return 1; }
  if (tts_vocoder(ctx, mel.data(), frames.data(), B, nullptr, noise_mode, audio.data())) return die(ctx, "vocoder");
  mark("vocoder");
  for (int c = 0; c < B; c++) nsamp.push_back((size_t)tts_vocoder_samples(frames[c]));
  } // !dry
  auto write_one = [&](const float *samples, int64_t ns, int gc, bool is_output) {
    const std::string path = is_output ? outputPath : outputPath + "." + std::to_string(gc) + ".wav";
    if (tts_write_wav(path.c_str(), samples, ns, 24000)) std::cerr << "Error opening output file." << std::endl;
    else if (is_output) std::cout << "WAV file saved successfully. :^)" << std::endl;
  };
  if (use_rccl) {
    auto bail = [&]() { fprintf(stderr, "rccl: %s\n", world.err.c_str()); return 1; };
    std::vector<char> g;
    if (kept_gc >= 0) { // every rank's winner: (score, global candidate, samples); the best one's audio goes to rank 0
      const double mine[3] = {kept_score, (double)kept_gc, (double)audio.size()};
      if (!world.all_gather(mine, sizeof mine, g)) return bail();
      const double *all = (const double *)g.data();
      int win = 0;
      for (int r = 1; r < nshards; r++)
        if (all[3 * r] > all[3 * win]) win = r;
      std::vector<int64_t> counts(nshards);
      for (int r = 0; r < nshards; r++) counts[r] = (int64_t)all[3 * r + 2];
      std::vector<float> got;
      if (!world.send_floats(audio.data(), counts, win, 0, got)) return bail();
      if (shard == 0) {
        printf("clvp: candidate %d kept (score %.5f)\n", (int)all[3 * win + 1], all[3 * win]);
        write_one(got.data(), (int64_t)got.size(), (int)all[3 * win + 1], true);
      }
    } else { // all candidates of all ranks: rank 0 writes <output> (candidate 0) and <output>.<c>.wav
      // sizes first: every rank's per-candidate sample counts (the stages' own numbers, so a --dry-run stand-in pairs exactly like a real run)
      std::vector<int64_t> mine_ns(nsamp.begin(), nsamp.end());
      if (!world.all_gather(mine_ns.data(), (size_t)B * 8, g)) return bail();
      const int64_t *alln = (const int64_t *)g.data();
      std::vector<int64_t> counts(nshards, 0);
      for (int r = 0; r < nshards; r++)
        for (int c = 0; c < B; c++) counts[r] += alln[r * B + c];
      for (int r = 0; r < nshards; r++) {
        std::vector<float> got;
        if (!world.send_floats(audio.data(), counts, r, 0, got)) return bail();
        if (shard == 0) {
          size_t off = 0;
          for (int c = 0; c < B; c++) {
            const int64_t ns = alln[r * B + c];
            write_one(got.data() + off, ns, r * B + c, r * B + c == 0);
            off += (size_t)ns;
          }
        }
      }
    }
  } else {
    size_t off = 0;
    for (int c = 0; c < B; c++) {
      size_t ns = nsamp[c];
      const int gc = kept_gc >= 0 ? kept_gc : (shard >= 0 ? shard * B_ar : 0) + c; // global candidate index
      // the re-ranked winner of a single process IS the output; a worker's winner waits for the parent's pick under its candidate name
      write_one(audio.data() + off, (int64_t)ns, gc, kept_gc >= 0 ? shard < 0 : gc == 0);
      off += ns;
    }
    if (kept_gc >= 0 && shard >= 0) {
      std::ofstream f(outputPath + ".shard" + std::to_string(shard) + ".score");
      f << kept_gc << " " << std::setprecision(17) << kept_score << "\n";
    }
  }
  mark("write");
  tts_destroy(ctx);
  mark("tts_destroy");
  return 0;
}
