// libsubgnn_hip.so: ABI version and error reporting.
#include "common.h"
#include <string.h>

static char g_last_error[256] = "";

void sgnn_set_last_error(hipError_t e) {
    strncpy(g_last_error, hipGetErrorString(e), sizeof(g_last_error) - 1);
    g_last_error[sizeof(g_last_error) - 1] = 0;
}

extern "C" int sgnn_abi_version(void) { return SGNN_ABI_VERSION; }
extern "C" const char* sgnn_last_error(void) { return g_last_error; }

// launches one empty kernel of every translation unit of the library on ``stream``: their code objects are loaded now, not
// in the middle of the first pass (common.h: SGNN_DEFINE_WARM).  Returns the number of units that failed to launch.
extern "C" int sgnn_warm_degree_sequence(void*);
extern "C" int sgnn_warm_graph_sets(void*);
extern "C" int sgnn_warm_samplers(void*);
extern "C" int sgnn_warm_similarity(void*);
extern "C" int sgnn_warm_dtw(void*);
extern "C" int sgnn_warm_embed(void*);
extern "C" int sgnn_warm_mpn(void*);
extern "C" int sgnn_warm_attention(void*);
extern "C" int sgnn_warm_lstm(void*);
extern "C" int sgnn_warm_probe(void*);
extern "C" int sgnn_warm_scatter(void*);
extern "C" int sgnn_warm_update(void*);
extern "C" int sgnn_warm_optim(void*);
extern "C" int sgnn_warm_readout(void*);
extern "C" int sgnn_warm_loss(void*);
extern "C" int sgnn_warm_head(void*);
extern "C" int sgnn_warm_up(void* stream)
{
    int bad = 0;
    bad += sgnn_warm_degree_sequence(stream) != 0;
    bad += sgnn_warm_graph_sets(stream) != 0;
    bad += sgnn_warm_samplers(stream) != 0;
    bad += sgnn_warm_similarity(stream) != 0;
    bad += sgnn_warm_dtw(stream) != 0;
    bad += sgnn_warm_embed(stream) != 0;
    bad += sgnn_warm_mpn(stream) != 0;
    bad += sgnn_warm_attention(stream) != 0;
    bad += sgnn_warm_lstm(stream) != 0;
    bad += sgnn_warm_probe(stream) != 0;
    bad += sgnn_warm_scatter(stream) != 0;
    bad += sgnn_warm_update(stream) != 0;
    bad += sgnn_warm_optim(stream) != 0;
    bad += sgnn_warm_readout(stream) != 0;
    bad += sgnn_warm_loss(stream) != 0;
    bad += sgnn_warm_head(stream) != 0;
    return bad == 0 ? SGNN_OK : SGNN_ERR_LAUNCH;
}
