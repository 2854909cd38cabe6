// Temporary: stages not implemented yet fail loudly.
#include "common.h"
namespace tts {
int voc_load(tts_ctx *c, const char *) { return fail(c, TTS_ERR_STATE, "vocoder stage not built yet"); }
void voc_free(VocState *) {}
int voc_run(tts_ctx *c, const float *, const int32_t *, int, const float *, int, float *) { return fail(c, TTS_ERR_STATE, "vocoder stage not built yet"); }
}
