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
// --steps <n> diffusion steps (default 80), --device <ordinal>.
#include "tortoise_mi355x.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

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
  int seed = 0, candidates = 1, steps = 80, device = 0, fixed_codes = 0;
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
  }
  tts_ctx *ctx = tts_create(device);
  if (!ctx) {
    fprintf(stderr, "tts_create(%d) failed: no HIP device (this engine has no CPU path)\n", device);
    return 1;
  }
  if (have_seed) tts_seed(ctx, (uint32_t)seed);
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
  if (tts_load_ar(ctx, (modelsDir + "/ggml-model.bin").c_str())) return die(ctx, "autoregressive_model_load");
  const int B = candidates;
  std::vector<int32_t> codes((size_t)B * 502), rows(B);
  std::vector<float> latents((size_t)B * 500 * 1024);
  int32_t nsteps = 0;
  if (tts_autoregressive(ctx, tokens.data(), n, voice.data(), B, fixed_codes > 0 ? fixed_codes : 500, fixed_codes > 0 ? TTS_AR_MASK_STOP : 0,
                         codes.data(), rows.data(), latents.data(), &nsteps))
    return die(ctx, "autoregressive");
  printf("tokens sampled: %d\n", nsteps);

  if (tts_load_diffusion(ctx, (modelsDir + "/ggml-diffusion-model.bin").c_str())) return die(ctx, "diffusion_model_load");
  size_t mel_total = 0, audio_total = 0;
  std::vector<int32_t> frames(B);
  for (int c = 0; c < B; c++) {
    frames[c] = tts_diffusion_frames(rows[c]);
    mel_total += (size_t)100 * frames[c];
    audio_total += (size_t)tts_vocoder_samples(frames[c]);
  }
  std::vector<float> mel(mel_total), audio(audio_total);
  // B == 1: the reference's exact RNG order (AR uniforms, x_T, per-step noise, vocoder noise)
  const int noise_mode = (B == 1) ? TTS_NOISE_REFERENCE : TTS_NOISE_DEVICE;
  if (tts_diffusion(ctx, latents.data(), rows.data(), B, steps, nullptr, noise_mode, mel.data())) return die(ctx, "diffusion");
  if (tts_load_vocoder(ctx, (modelsDir + "/ggml-vocoder-model.bin").c_str())) return die(ctx, "vocoder_model_load");
  if (tts_vocoder(ctx, mel.data(), frames.data(), B, nullptr, noise_mode, audio.data())) return die(ctx, "vocoder");
  size_t off = 0;
  for (int c = 0; c < B; c++) {
    size_t ns = (size_t)tts_vocoder_samples(frames[c]);
    std::string path = (c == 0) ? outputPath : outputPath + "." + std::to_string(c) + ".wav";
    if (tts_write_wav(path.c_str(), audio.data() + off, (int64_t)ns, 24000)) {
      std::cerr << "Error opening output file." << std::endl;
    } else if (c == 0) {
      std::cout << "WAV file saved successfully. :^)" << std::endl;
    }
    off += ns;
  }
  tts_destroy(ctx);
  return 0;
}
