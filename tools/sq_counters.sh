# SQ / TCP / TCC counters of the hot kernels on one 3x3x512x512 window pass per step (tools/pass_time.py: headline parameters,
# then the HT step with tau_2D = bior for BASELINE configs[3]'s k_group_bior16_haar), summarised into ONE tracked file.
# Counters only (own runs, no tracing); kernel durations from a separate --kernel-trace --stats run.  On the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/sq_counters.sh r04_a'      ->  gpurun_out/<tag>_sq_counters.txt  (copy to profiles/)
tag=${1:-r04_x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sqc_$tag; rm -rf $out; mkdir -p $out
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
)
for cfg in "id" "bior"; do
  i=0
  for s in "${sets[@]}"; do
    rocprofv3 --pmc $s --output-format csv -d $out/$cfg/p$i -- python3 tools/pass_time.py 1 512 $cfg > /dev/null 2>&1
    i=$((i+1))
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$cfg/stats -o s -- python3 tools/pass_time.py 5 512 $cfg > /dev/null 2>&1
done
python3 - "$tag" "$out" <<'PY' > gpurun_out/${tag}_sq_counters.txt
import csv, glob, collections, sys
tag, out = sys.argv[1], sys.argv[2]
KERNELS = [("k_bm_scan2<16", "id"), ("k_bm_scan2<8", "id"), ("k_self_select", "id"), ("k_stereo_argmin3", "id"), ("k_group_id_haar", "id"), ("k_group_dct8w3", "id"),
           ("k_aggregate<false", "id"), ("k_aggregate<true", "id"), ("k_group_bior16_haar", "bior")]
print(f"# {tag}: SQ / TCP / TCC counters per launch, one 3x3x512x512 (560^2 padded) window pass per step, README parameters (k_group_bior16_haar: HT step with")
print("# tau_2D = bior, BASELINE configs[3]).  Collected by tools/sq_counters.sh: rocprofv3 --pmc <set> in separate counter-only runs of tools/pass_time.py 1")
print("# (the first pass of a context, one launch per kernel and step), durations from a --kernel-trace --stats run of tools/pass_time.py 5.")
print("# Derived figures (stated so that they can be recomputed from the raw values below):")
print("#   issue[X]   = SQ_ACTIVE_INST_X / SQ_WAVE_CYCLES: share of a resident wave's cycles with an instruction of class X in flight")
print("#   wait       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES: ... waiting for an instruction to issue / return")
print("#   occupancy  = SQ_WAVE_CYCLES / (4 x SQ_BUSY_CYCLES): resident waves per SIMD while the shader engines are busy (SQ_BUSY_CYCLES is summed over the")
print("#                SQs of the device like every other counter here, so the ratio is per SQ)")
print("#   lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE: share of LDS-active cycles spent in bank conflicts")
print("#   l2_hit     = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)")
for kern, cfg in KERNELS:
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(f"{out}/{cfg}/p*/**/*counter_collection.csv", recursive=True):
        p = f.split("/p")[1][0]
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[(p, r["Counter_Name"])].add(r["Dispatch_Id"])
    if not acc:
        print(f"\n{kern}: no launches recorded"); continue
    per = {c: v / max(1, len(n[[k for k in n if k[1] == c][0]])) for c, v in acc.items()}
    dur = None
    for f in glob.glob(f"{out}/{cfg}/stats/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Name"]: dur = float(r["AverageNs"]) / 1e3
    g = lambda c: per.get(c, float("nan"))
    print(f"\n{kern}  ({cfg} configuration; average launch {dur:.1f} us)" if dur else f"\n{kern}")
    wc = g("SQ_WAVE_CYCLES")
    print("  issue: VALU %.2f  LDS %.2f  VMEM %.2f  scalar %.2f  any %.2f;  wait %.2f;  occupancy %.2f waves/SIMD;  lds_conflict %.2f;  l2_hit %.2f" % (
        g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_ACTIVE_INST_LDS") / wc, g("SQ_ACTIVE_INST_VMEM") / wc, g("SQ_ACTIVE_INST_SCA") / wc, g("SQ_ACTIVE_INST_ANY") / wc,
        g("SQ_WAIT_INST_ANY") / wc, wc / (4 * g("SQ_BUSY_CYCLES")), g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")),
        g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
    print("  instructions per launch: VALU %.3g  SALU %.3g  LDS %.3g  VMEM read %.3g  VMEM write %.3g;  waves %.3g  (VALU per wave %.0f)" % (
        g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_WAVES"), g("SQ_INSTS_VALU") / max(1.0, g("SQ_WAVES"))))
    print("  raw: " + "  ".join(f"{c}={per[c]:.4g}" for c in sorted(per)))
PY
cat gpurun_out/${tag}_sq_counters.txt | cut -c1-260
rm -rf $out
