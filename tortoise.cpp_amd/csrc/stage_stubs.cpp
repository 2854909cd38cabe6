// Temporary: stages not implemented yet fail loudly.
#include "common.h"
namespace tts {
int diff_load(tts_ctx *c, const char *) { return fail(c, TTS_ERR_STATE, "diffusion stage not built yet"); }
void diff_free(DiffState *) {}
int diff_layers(const tts_ctx *) { return 0; }
int diff_forward(tts_ctx *c, const float *, int, const float *, int, int, float *) { return fail(c, TTS_ERR_STATE, "diffusion stage not built yet"); }
int diff_sample(tts_ctx *c, const float *, const int32_t *, int, int, const float *, int, float *) { return fail(c, TTS_ERR_STATE, "diffusion stage not built yet"); }
int voc_load(tts_ctx *c, const char *) { return fail(c, TTS_ERR_STATE, "vocoder stage not built yet"); }
void voc_free(VocState *) {}
int voc_run(tts_ctx *c, const float *, const int32_t *, int, const float *, int, float *) { return fail(c, TTS_ERR_STATE, "vocoder stage not built yet"); }
}
