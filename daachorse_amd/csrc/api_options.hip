// C ABI (include/daachorse_amd.h), part 5: options — process-wide (daac_set_option) and per handle (daac_pma_set_option).
// The list itself is DAAC_OPTIONS in api_internal.hpp: every name there is one that tests/, tools/ or bench.py set.
#include "api_internal.hpp"

extern "C" {

// user-facing option name -> index into Options; `value` is clamped where the option has a range; `upload` = read when tables are laid out
static bool option_slot(const std::string &n, int64_t &value, int &id, bool &upload) {
#define X(NAME, DEF, UP) if (n == #NAME) { id = OPT_##NAME; upload = (UP) != 0; goto found; }
    DAAC_OPTIONS(X)
#undef X
    if (n == "gram3_tail") { id = OPT_gram_tail; upload = false; goto found; }   // (its name while the kernel was gram3_kernels.hip)
    return false;
found:
    if (id == OPT_find3_window) value = std::min<int64_t>(1ll << 30, std::max<int64_t>(8192, value));
    if (id == OPT_workspace_keep) value = std::max<int64_t>(0, value);
    if (id == OPT_gram_version && value == 3) value = 4;   // ABI 4's name of the dedicated `.count()` kernel
    return true;
}

daac_status daac_set_option(const char *name, int64_t value) {
    if (!name) { set_error("null option name"); return DAAC_ERR_INVALID_ARGUMENT; }
    int id = 0;
    bool upload = false;
    if (!option_slot(name, value, id, upload)) { set_error(std::string("unknown option: ") + name); return DAAC_ERR_INVALID_ARGUMENT; }
    g_opt.v[id].store(value);
    return DAAC_OK;
}

// The same option for ONE handle: overrides the process-wide value for every scan, iterator and stream of `pma`.  `unset` != 0 takes the
// override away.  `pool` / `pool_keep` (the device's allocator) have no per-handle meaning: status 1.  An option that is read when the
// tables are laid out (gram_lds_budget, lds_budget, pfx, left3, char_map_lds, char_row_lds, dense_depth, rows_share_pct, gram_rank_in_lds)
// can only be set BEFORE the handle's first upload: afterwards the tables keep their layout, so the call changes nothing and says so
// (DAAC_ERR_UNSUPPORTED) instead of returning OK for an override that would never take effect (round-5 advisor).
daac_status daac_pma_set_option(daac_pma *pma, const char *name, int64_t value, int unset) {
    if (!pma || !name) { set_error("null argument"); return DAAC_ERR_INVALID_ARGUMENT; }
    int id = 0;
    bool upload = false;
    if (!option_slot(name, value, id, upload)) { set_error(std::string("unknown option: ") + name); return DAAC_ERR_INVALID_ARGUMENT; }
    if (id == OPT_pool || id == OPT_pool_keep) { set_error("pool / pool_keep are properties of the device's allocator, not of a handle"); return DAAC_ERR_INVALID_ARGUMENT; }
    if (upload) {
        std::lock_guard<std::mutex> g(pma->mu);
        if (!pma->dev.empty()) {
            set_error(std::string("option ") + name + " is read when the handle's tables are laid out; this handle already has tables on a device (set it before the first daac_pma_upload / scan)");
            return DAAC_ERR_UNSUPPORTED;
        }
    }
    OptionOverrides &o = pma->opt_ov;
    if (unset) {
        o.mask.fetch_and(~(1ull << id), std::memory_order_acq_rel);
    } else {
        o.v[id].store(value, std::memory_order_relaxed);
        o.mask.fetch_or(1ull << id, std::memory_order_acq_rel);
    }
    return DAAC_OK;
}

}  // extern "C"
