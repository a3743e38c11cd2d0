// C ABI (include/daachorse_amd.h), part 5: options — process-wide (daac_set_option) and per handle (daac_pma_set_option).
#include "api_internal.hpp"

extern "C" {

// user-facing option name -> {field of Options (the name OPT() looks up), the process-wide atomic}; `value` is clamped where the option has a range
static bool option_slot(const std::string &n, int64_t &value, const char *&field, std::atomic<int64_t> *&slot) {
#define SLOT(NAME) if (n == #NAME) { field = #NAME; slot = &g_opt.NAME; return true; }
    SLOT(seg_bytes) SLOT(lds_budget) SLOT(dense_depth) SLOT(rows_share_pct) SLOT(blocks_per_cu) SLOT(threads) SLOT(iter_window) SLOT(max_result_bytes)
    SLOT(gram_lds_budget) SLOT(gram_region) SLOT(gram_slab) SLOT(gram_dense) SLOT(gram_ppl) SLOT(gram_version) SLOT(gram2_dpp) SLOT(gram2_rfull)
    SLOT(gram4_arith) SLOT(gram3_tail) SLOT(pfx) SLOT(pfx_probe) SLOT(find3) SLOT(left3) SLOT(select_emit) SLOT(emit) SLOT(emit_stagger) SLOT(emit_v3_lds)
    SLOT(emit_rec_per_kib) SLOT(gram_rank_in_lds) SLOT(restart_chain) SLOT(restart_bpc) SLOT(chain_rounds) SLOT(overlap_micro) SLOT(pool) SLOT(pool_keep)
    SLOT(char_map_lds) SLOT(char_row_lds)
#undef SLOT
    if (n == "gram_tail") { field = "gram3_tail"; slot = &g_opt.gram3_tail; return true; }
    if (n == "find3_window") { value = std::min<int64_t>(1ll << 30, std::max<int64_t>(8192, value)); field = "find3_window"; slot = &g_opt.find3_window; return true; }
    if (n == "workspace_keep") { value = std::max<int64_t>(0, value); field = "workspace_keep"; slot = &g_opt.workspace_keep; return true; }
    // options of engines that left the library (the TIERED chain walkers, the round-3 COUNT + WRITE emitter): accepted, nothing left to steer
    if (n == "restart_tier" || n == "emit_tiles" || n == "emit_rec_cap" || n == "emit_version") { field = nullptr; slot = nullptr; return true; }
    return false;
}

daac_status daac_set_option(const char *name, int64_t value) {
    if (!name) { set_error("null option name"); return DAAC_ERR_INVALID_ARGUMENT; }
    const char *field = nullptr;
    std::atomic<int64_t> *slot = nullptr;
    if (!option_slot(name, value, field, slot)) { set_error(std::string("unknown option: ") + name); return DAAC_ERR_INVALID_ARGUMENT; }
    if (slot) slot->store(value);
    return DAAC_OK;
}

// The same option for ONE handle: overrides the process-wide value for every scan, iterator and stream of `pma` (and, for the options read
// when the tables are laid out — gram_lds_budget, pfx, left3, lds_budget, char_map_lds ... — for its next daac_pma_upload).  `unset` != 0 takes
// the override away.  `pool` / `pool_keep` (the device's allocator) have no per-handle meaning: status 1.
daac_status daac_pma_set_option(daac_pma *pma, const char *name, int64_t value, int unset) {
    if (!pma || !name) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    const char *field = nullptr;
    std::atomic<int64_t> *slot = nullptr;
    if (!option_slot(name, value, field, slot)) { set_error(std::string("unknown option: ") + name); return DAAC_ERR_INVALID_ARGUMENT; }
    if (slot == &g_opt.pool || slot == &g_opt.pool_keep) { set_error("pool / pool_keep are properties of the device's allocator, not of a handle"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (!field) return DAAC_OK;
    std::lock_guard<std::mutex> g(pma->opt_mu);
    if (unset) pma->opt_ov.erase(field); else pma->opt_ov[field] = value;
    pma->opt_n.store(static_cast<int>(pma->opt_ov.size()));
    return DAAC_OK;
}

}  // extern "C"

