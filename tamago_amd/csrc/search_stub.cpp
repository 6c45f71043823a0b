// TEMPORARY: search entry points under construction (replaced by search.hip).
#include "common.h"
#define NI(name, ...) extern "C" int name(__VA_ARGS__) { return tg::fail(TG_ERR_STATE, #name ": not built yet"); }
NI(tg_search_create, const tg_search_config *, tg_search **)
NI(tg_search_destroy, tg_search *)
NI(tg_search_set_zobrist, tg_search *, const uint64_t *, size_t)
NI(tg_search_set_root, tg_search *, int, const tg_root_position *)
NI(tg_search_set_rng, tg_search *, const double *, size_t, size_t)
NI(tg_search_rng_consumed, tg_search *, int64_t *)
NI(tg_search_select_puct, tg_search *, int, float *, int32_t *, void *)
NI(tg_search_root_planes, tg_search *, float *, void *)
NI(tg_search_backup, tg_search *, const float *, const float *, int, void *)
NI(tg_search_read_node, tg_search *, int, int, int32_t *, int32_t *, int32_t *, int32_t *, int32_t *, int32_t *, double *, double *, double *, float *, float *)
NI(tg_search_num_nodes, tg_search *, int32_t *)
