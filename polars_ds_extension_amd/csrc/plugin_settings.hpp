// plugin_settings.hpp -- every environment variable the plugin layer looks at, read ONCE (the first expression call of the process)
// Part of the one translation unit plugin.cpp (included there, inside its anonymous namespace, before the other plugin_*.hpp).
// A host that changes the environment afterwards (the tests do) calls pds_plugin_reload_settings() -- not while expressions run.
#pragma once

struct PluginSettings {
    int device = 0;                               // PDS_DEVICE: the device of the per-thread contexts
    std::string devices;                          // PDS_DEVICES: "0,1,..." or "all" -- devices of the sliced `pl_lr_by` route (default: `device`)
    int by_key_contexts = 2;                      // PDS_BY_KEY_CONTEXTS: contexts per device on that route
    int64_t by_key_multi_min_rows = (int64_t)1 << 22;  // PDS_BY_KEY_MULTI_MIN_ROWS: frames from this many rows take it (0: never)
    int by_key_slices = 0;                        // PDS_BY_KEY_SLICES: slice count (0: chosen from the frame)
    int coalesce = 1;                             // PDS_PLUGIN_COALESCE: 0 per-call launches, 1 the coalescing queue, 2 the queue's batch path without queueing
    bool reference_quirks = false;                // PDS_REFERENCE_QUIRKS=1: the reference's two result-assembly accidents, as it has them
    size_t pinned_cache_bytes = (size_t)4 << 30;  // PDS_PLUGIN_PINNED_CACHE_MB: page-locked result blocks kept between calls
    bool pinned_results = true;                   // PDS_PLUGIN_PINNED_RESULTS=0: pageable result storage
    // the two CONTEXT options (pds_ctx_set_option) as the plugin layer's contexts get them: a context latches the environment when it
    // is created, the plugin's contexts live for the process -- so a reload re-applies these to every context at its next use
    bool keyed_sort = false;                      // PDS_KEYED_SORT=1: the determinism switch of the keyed route
    bool wide_f32_native = false;                 // PDS_WIDE_F32_NATIVE=1

    static PluginSettings from_env() {
        PluginSettings s;
        auto env = [](const char* name) { return std::getenv(name); };  // (the plugin layer's only look at the environment)
        if (const char* e = env("PDS_DEVICE")) s.device = std::atoi(e);
        if (const char* e = env("PDS_DEVICES")) s.devices = e;
        if (const char* e = env("PDS_BY_KEY_CONTEXTS")) s.by_key_contexts = std::max(1, std::atoi(e));
        if (const char* e = env("PDS_BY_KEY_MULTI_MIN_ROWS")) s.by_key_multi_min_rows = (int64_t)std::atoll(e);
        if (const char* e = env("PDS_BY_KEY_SLICES")) s.by_key_slices = std::atoi(e);
        if (const char* e = env("PDS_PLUGIN_COALESCE")) s.coalesce = (e[0] == '0') ? 0 : (e[0] == '2' ? 2 : 1);
        if (const char* e = env("PDS_REFERENCE_QUIRKS")) s.reference_quirks = e[0] == '1';
        if (const char* e = env("PDS_PLUGIN_PINNED_CACHE_MB")) s.pinned_cache_bytes = (size_t)std::max<long long>(0, std::atoll(e)) << 20;
        if (const char* e = env("PDS_PLUGIN_PINNED_RESULTS")) s.pinned_results = e[0] != '0';
        if (const char* e = env("PDS_KEYED_SORT")) s.keyed_sort = e[0] == '1';
        if (const char* e = env("PDS_WIDE_F32_NATIVE")) s.wide_f32_native = e[0] == '1';
        return s;
    }
};
std::atomic<int> g_settings_epoch{1};  // bumped by pds_plugin_reload_settings: contexts re-apply the context options when they see a new one
PluginSettings& settings() {
    static PluginSettings* s = new PluginSettings(PluginSettings::from_env());  // (never destroyed: results may outlive static destruction)
    return *s;
}
