/*
 * lfbm5d_options.h -- the run-time options of a context (round 6).
 *
 * Rounds 1-5 read 28 LFBM5D_* environment variables inside the library, many of them on every call: process-global, not
 * thread-safe, invisible in the C-ABI.  Now every knob is a field of this struct, which belongs to a context:
 *   - lfbm5d_create fills it ONCE from the environment (options_from_env: the library's only getenv site), so a program that
 *     exports LFBM5D_LANES=3 before creating its context behaves as before;
 *   - lfbm5d_set_option / lfbm5d_get_option (include/lfbm5d.h) change / read it afterwards, per context;
 *   - the kernel-generation selectors (test hooks: which of two implementations of the same arithmetic runs) travel to the
 *     launch functions as a bit mask in their argument structs (GroupArgs::opt, AggArgs::opt, ScanArgs::opt).
 * Keys are the lower-case field names below; the old variable names (LFBM5D_LANES ...) are accepted as aliases.
 */
#ifndef LFBM5D_OPTIONS_H
#define LFBM5D_OPTIONS_H

#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>

namespace lfbm5d {

/* bits of GroupArgs::opt / AggArgs::opt / ScanArgs::opt */
enum : unsigned {
    kOptScanV1 = 1u << 0,           /* round 2's table kernel (one wave per table) instead of the ring-sharing one */
    kOptScanAny = 1u << 1,          /* the any-patch-size table kernel for every patch size */
    kOptScanFullTables = 1u << 2,   /* ring-sharing kernel with whole disparity tables in memory (second-generation layout) */
    kOptDct8wV2 = 1u << 3,          /* Wiener 8x8 DCT: k_group_dct8w2 for every configuration */
    kOptGroupGeneric = 1u << 4,     /* every configuration through the generic LDS group kernel */
    kOptNoSaKernels = 1u << 5,      /* windows with an empty SAI on the ordinary kernels (call form of the shape-adaptive transform) */
    kOptNoSlabKernel = 1u << 6,     /* stacks beyond the LDS on the general kernel's HBM slices */
    kOptWideNoSplit = 1u << 7,      /* wide-window kernel without its SPLIT form */
    kOptAgg64Bit = 1u << 8,         /* aggregation: 64-bit gather addresses whatever the size of filt */
    kOptAggScalarScan = 1u << 9,    /* aggregation: one candidate per lane in the scan */
    kOptSubsetListHost = 1u << 10,  /* subset passes: reference list built on the host */
    kOptSubsetScanV1 = 1u << 11,    /* subset passes: round 2's table kernel through the position map */
    kOptFiltGroupMajor = 1u << 12,  /* wide windows (11 x 11 SAIs and more): filtered patches group-major like the 3 x 3 windows' (rounds 1-5) instead of SAI-major */
};

struct Options {
    /* user knobs */
    int lanes = 2;                  /* window lanes of the graph form (1..8) */
    int emulate_world = 0;          /* > 1: that many ranks played on this GPU (tests, bench.py --emulate) */
    int max_windows = 0;            /* > 0: stop each step after that many windows (bench.py's CPU comparison) */
    int fused = 1;                  /* 0: lfbm5d_denoise_* runs the two calls one after the other */
    int step_sharding = 0;          /* 0: window graph, 1: rows (every pass sharded by reference rows + all-reduce), 2: blocks */
    int data_driven_schedule = 0;   /* 1: windows chosen from the device's den == 0 counts like the reference, not planned from the mask */
    int host_blocking = 0;          /* 1: host seam with four blocking copies of the light field (rounds 1-4) */
    int band_mb = 0;                /* > 0: cap of the filt buffer in MiB (passes beyond it run band by band) */
    int bm3d_lanes = 3;             /* SAIs of LFBM3Ddenoising processed concurrently */
    int scan_lds_cap = 0;           /* > 0: LDS bytes the first-generation table kernel may use */
    int force_redo = 0;             /* test hook: treat the graph as incomplete once (exercises the sequential redo) */
    int spatial_bands = 1;          /* 0: chosen by lfbm5d_auto_bands; S > 1, several ranks, lfbm5d_denoise_*: S teams of ranks, each denoises a horizontal band of every SAI (+ halo) on its own
                                     * window graph; NOT bit-identical to one GPU, PSNR within 1e-3 dB (lfbm5d_steps.hip) */
    int band_halo = 0;              /* rows of halo of a band; 0: nSim + nDisp + k of the wider step */
    unsigned kernels = 0;           /* kOpt* bits: kernel-generation selectors */
};

struct OptionKey { const char* key; const char* env; int Options::* field; unsigned bit; };

inline const OptionKey* option_keys(size_t* n) {
    static const OptionKey keys[] = {
        {"lanes", "LFBM5D_LANES", &Options::lanes, 0}, {"emulate_world", "LFBM5D_EMULATE_WORLD", &Options::emulate_world, 0},
        {"max_windows", "LFBM5D_MAX_WINDOWS", &Options::max_windows, 0}, {"fused", "LFBM5D_FUSED", &Options::fused, 0},
        {"step_sharding", "LFBM5D_STEP_SHARDING", &Options::step_sharding, 0},
        {"data_driven_schedule", "LFBM5D_DATA_DRIVEN_SCHEDULE", &Options::data_driven_schedule, 0},
        {"host_blocking", "LFBM5D_HOST_BLOCKING", &Options::host_blocking, 0}, {"band_mb", "LFBM5D_BAND_MB", &Options::band_mb, 0},
        {"bm3d_lanes", "LFBM5D_BM3D_LANES", &Options::bm3d_lanes, 0}, {"scan_lds_cap", "LFBM5D_SCAN_LDS_CAP", &Options::scan_lds_cap, 0},
        {"force_redo", "LFBM5D_FORCE_REDO", &Options::force_redo, 0}, {"spatial_bands", "LFBM5D_SPATIAL_BANDS", &Options::spatial_bands, 0},
        {"band_halo", "LFBM5D_BAND_HALO", &Options::band_halo, 0},
        {"scan_v1", "LFBM5D_SCAN_V1", nullptr, kOptScanV1}, {"scan_any", "LFBM5D_SCAN_ANY", nullptr, kOptScanAny},
        {"scan_full_tables", "LFBM5D_SCAN_FULL_TABLES", nullptr, kOptScanFullTables}, {"dct8w_v2", "LFBM5D_DCT8W_V2", nullptr, kOptDct8wV2},
        {"group_generic", "LFBM5D_GROUP_GENERIC", nullptr, kOptGroupGeneric}, {"no_sa_kernels", "LFBM5D_NO_SA_KERNELS", nullptr, kOptNoSaKernels},
        {"no_slab_kernel", "LFBM5D_NO_SLAB_KERNEL", nullptr, kOptNoSlabKernel}, {"wide_nosplit", "LFBM5D_WIDE_NOSPLIT", nullptr, kOptWideNoSplit},
        {"agg_64bit", "LFBM5D_AGG_64BIT", nullptr, kOptAgg64Bit}, {"agg_scalar_scan", "LFBM5D_AGG_SCALAR_SCAN", nullptr, kOptAggScalarScan},
        {"subset_list_host", "LFBM5D_SUBSET_LIST_HOST", nullptr, kOptSubsetListHost}, {"subset_scan_v1", "LFBM5D_SUBSET_SCAN_V1", nullptr, kOptSubsetScanV1},
        {"filt_group_major", "LFBM5D_FILT_GROUP_MAJOR", nullptr, kOptFiltGroupMajor},
    };
    *n = sizeof(keys) / sizeof(keys[0]);
    return keys;
}

/* value of an option as the environment spelled it: integers; "rows" / "blocks" for step_sharding; for the flags any
 * non-empty text other than "0" means on (the old variables were tested for presence); NULL / "" resets to the default */
inline bool option_set(Options& o, const char* key, const char* value) {
    size_t n; const OptionKey* keys = option_keys(&n);
    static const Options defaults;
    for (size_t i = 0; i < n; i++) {
        if (std::strcmp(key, keys[i].key) != 0 && std::strcmp(key, keys[i].env) != 0) continue;
        const bool unset = value == nullptr || value[0] == 0;
        if (keys[i].field) {
            int v = defaults.*(keys[i].field);
            if (!unset) {
                if (keys[i].field == &Options::step_sharding) v = std::strcmp(value, "rows") == 0 ? 1 : std::strcmp(value, "blocks") == 0 ? 2 : std::atoi(value);
                else if (keys[i].field == &Options::data_driven_schedule || keys[i].field == &Options::host_blocking || keys[i].field == &Options::force_redo)
                    v = std::strcmp(value, "0") != 0;   /* presence flags of rounds 1-5 */
                else v = std::atoi(value);
            }
            o.*(keys[i].field) = v;
        } else {
            if (!unset && std::strcmp(value, "0") != 0) o.kernels |= keys[i].bit; else o.kernels &= ~keys[i].bit;
        }
        return true;
    }
    return false;
}
inline bool option_get(const Options& o, const char* key, std::string& out) {
    size_t n; const OptionKey* keys = option_keys(&n);
    for (size_t i = 0; i < n; i++) {
        if (std::strcmp(key, keys[i].key) != 0 && std::strcmp(key, keys[i].env) != 0) continue;
        if (keys[i].field == &Options::step_sharding) out = o.step_sharding == 1 ? "rows" : o.step_sharding == 2 ? "blocks" : "0";
        else if (keys[i].field) out = std::to_string(o.*(keys[i].field));
        else out = (o.kernels & keys[i].bit) ? "1" : "0";
        return true;
    }
    return false;
}
/* the library's only read of the environment: once per context, at lfbm5d_create */
inline void options_from_env(Options& o) {
    size_t n; const OptionKey* keys = option_keys(&n);
    for (size_t i = 0; i < n; i++)
        if (const char* e = std::getenv(keys[i].env)) {
            /* a variable that is present but empty counted as "set" for the presence flags of rounds 1-5 */
            const bool flag = !keys[i].field || keys[i].field == &Options::data_driven_schedule || keys[i].field == &Options::host_blocking ||
                              keys[i].field == &Options::force_redo;
            if (e[0]) (void)option_set(o, keys[i].key, e); else if (flag) (void)option_set(o, keys[i].key, "1");
        }
}

} /* namespace lfbm5d */
#endif
