"""Build container: fill the round-6 figures into DESIGN.md (placeholders @@name@@; the template is kept as
docs/DESIGN.template.md) from gpurun_out/r6/ (what tools/collect_r6.sh left; tools/publish_r6.sh copies the same files into
profiles/).  `--table` prints the results table (README.md).
    python tools/r6_fill_docs.py [--table]"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", "r6")


def L(name):
    return json.loads(open(f"{R}/bench_{name}.json").read().strip().splitlines()[-1])


def stats(name, kernel):
    rows = list(csv.DictReader(open(glob.glob(f"{R}/prof_{name}/*kernel_stats.csv")[0])))
    c = sum(int(r["Calls"]) for r in rows if r["Name"].replace("void ", "").startswith(kernel))
    t = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].replace("void ", "").startswith(kernel))
    return c / 25.0, t / 25.0 / 1e6, t / max(c, 1) / 1e3          # launches per step, ms per step, us per launch


ROWS = [("2 (fp32, the metric's configuration; r6_bench_default.json)", "default"),
        ("2, bf16 storage (r6_bench_cfg2_bf16.json)", "cfg2_bf16"),
        ("2 with BatchNormReLU units (`cfg2-bn`, layer-by-layer path; r6_bench_cfg2_bn.json)", "cfg2_bn"),
        ("3 crop + mask branch only (64 synthetic boxes known before the forward), fp32 (r6_bench_cfg3.json)", "cfg3"),
        ("3 crop + mask branch only, bf16 storage (r6_bench_cfg3_bf16.json)", "cfg3_bf16"),
        ("3-rpn: a STAND-IN RPN inside the step (one anchor level, 2 x 32 stack, inside anchors, 64 kept), fp32 (r6_bench_cfg3rpn.json)", "cfg3rpn"),
        ("3-rpn, bf16 storage (r6_bench_cfg3rpn_bf16.json)", "cfg3rpn_bf16"),
        ("**ref-crop-rpn: the reference's RPN shape** (12 crops of 128x128x64, plan 32-112, two anchor levels, 5 x 128 / 5 x 256 stacks, "
         "256 kept, 24 per sample to the mask head), fp32 (r6_bench_ref_crop_rpn.json)", "ref_crop_rpn"),
        ("ref-crop-rpn, bf16 storage (r6_bench_ref_crop_rpn_bf16.json)", "ref_crop_rpn_bf16"),
        ("5 shape (600 k voxels, 5 levels to 512), fp32 (r6_bench_cfg5_fp32.json)", "cfg5_fp32"),
        ("5 shape, bf16 storage (r6_bench_cfg5_bf16.json)", "cfg5_bf16"),
        ("`ref`: the reference's own plan 32-48-64-80-96-112, cfg-2 scene (r6_bench_ref.json)", "ref"),
        ("`ref-crop`: that plan on the reference's training batch (r6_bench_ref_crop.json)", "ref_crop")]


def table():
    out = ["| Config | ms/step | active-voxels/s (fwd+bwd, rulebooks included) | forward only, ms (in-process / fresh process) | "
           "peak HBM, training / evaluation-only process (GB) | dominant kernel vs roofline (HIP events, live) |", "|---|---|---|---|---|---|"]
    for label, name in ROWS:
        d = L(name); r = d["roofline"]; fo = d.get("forward_only", {}); fr = fo.get("fresh_process", {}) or {}
        roof = f"`{r['kernel']}` {r['achieved']:.1f} {r['unit']} = {r['frac']:.3f} of {r['peak']:g}"
        if r.get("traffic"):
            roof += f"; PMC {r['traffic'] / 1e6:.0f} MB per launch vs {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic"
        frs = f"{fr['ms_per_step']:.2f}" if fr.get("ms_per_step") else "--"
        frp = f"{fr['peak_hbm_bytes'] / 1e9:.2f}" if fr.get("peak_hbm_bytes") else "--"
        out.append(f"| {label} | **{d['ms_per_step']:.2f}** | {d['value'] / 1e6:.1f} M | {d.get('forward_only_ms', float('nan')):.2f} / {frs} | "
                   f"{d['peak_hbm_bytes'] / 1e9:.2f} / {frp} | {roof} |")
    d, e, f = L("n2_gloo_one_gpu"), L("n2_gloo_cfg3rpn_bf16"), L("rccl_one_rank")
    out.append(f"| 2 / 3-rpn-bf16, two gloo ranks sharing ONE GPU (rehearsal of the N > 1 code path, not a scaling point) | {d['ms_per_step']:.1f} / "
               f"{e['ms_per_step']:.1f} | -- | | | -- |")
    out.append(f"| 2, ONE rank with the NCCL process group forced (rehearsal of the RCCL calls) | {f['ms_per_step']:.2f} | {f['value'] / 1e6:.1f} M | | | -- |")
    out.append("| 2 / 3 / 4 / 5 at 2, 4, 8 GPUs | **no scaling curve exists**: this pool hands out one GPU per call; the RCCL path has run with one "
               "rank only | | | | |")
    return "\n".join(out)


def values():
    d = L("default"); r = d["roofline"]; cb = d["cpu_baseline"]
    n_c, ms_c, us_c = stats("cfg2", "k_conv_ts")
    n_w, ms_w, us_w = stats("cfg2", "k_wgrad_direct")
    b = L("cfg2_bf16"); rb = b["roofline"]
    return {"cfg2_ms": f"{d['ms_per_step']:.2f}", "cfg2_mv": f"{d['value'] / 1e6:.1f}", "conv_ms": f"{ms_c:.2f}", "conv_tf": f"{r['achieved']:.1f}",
            "conv_frac": f"{r['frac']:.3f}", "conv_us": f"{us_c:.1f}",
            "conv_frac_prof": f"{r['algorithmic_gflop_per_launch'] * 1e9 / (us_c * 1e-6) / 1e12 / 157.3:.3f}",
            "conv_traffic": f"{(r.get('traffic') or 0) / 1e6:.0f}", "wgrad_ms": f"{ms_w:.2f}",
            "cfg2_bf16_ms": f"{b['ms_per_step']:.2f}", "cfg5_bf16_ms": f"{L('cfg5_bf16')['ms_per_step']:.2f}",
            "tb_gbs": f"{rb['achieved']:.0f}", "tb_frac": f"{rb['frac']:.3f}", "tb_traffic": f"{(rb.get('traffic') or 0) / 1e6:.0f}",
            "cfg3_ms": f"{L('cfg3')['ms_per_step']:.2f}", "cfg3_bf16_ms": f"{L('cfg3_bf16')['ms_per_step']:.2f}",
            "cfg3rpn_ms": f"{L('cfg3rpn')['ms_per_step']:.2f}", "cfg3rpn_bf16_ms": f"{L('cfg3rpn_bf16')['ms_per_step']:.2f}",
            "refrpn_ms": f"{L('ref_crop_rpn')['ms_per_step']:.1f}", "refrpn_bf16_ms": f"{L('ref_crop_rpn_bf16')['ms_per_step']:.1f}",
            "refrpn_fo": f"{L('ref_crop_rpn')['forward_only_ms']:.1f}", "refrpn_bf16_fo": f"{L('ref_crop_rpn_bf16')['forward_only_ms']:.1f}",
            "cpu16": f"{cb['value'] / 1e3:.0f}", "cpu1": f"{cb['single_thread_value'] / 1e3:.1f}", "index_ms": f"{d['index_build_ms']:.2f}",
            "x16": f"{d['value'] / cb['value']:.0f}", "x1": f"{d['value'] / cb['single_thread_value']:.0f}",
            "noprefetch": f"{d['ms_per_step_no_prefetch']:.2f}", "dropin": f"{d['dropin']['ms_per_step']:.2f}",
            "dropin_pf": f"{d['dropin']['ms_per_step_index_prefetching']:.2f}",
            "dropin_bf16": f"{d['dropin']['bf16']['ms_per_step']:.2f}",
            "dropin_bf16_pf": f"{d['dropin']['bf16']['ms_per_step_index_prefetching']:.2f}",
            "changing": f"{d['changing_scenes']['ms_per_step']:.2f}",
            "TABLE": table()}


if __name__ == "__main__":
    v = values()
    p = os.path.join(ROOT, "DESIGN.md")
    tpl = os.path.join(ROOT, "docs", "DESIGN.template.md")
    s = open(p).read()
    if "@@" in s:
        open(tpl, "w").write(s)                       # first run: keep the template
    elif os.path.exists(tpl):
        s = open(tpl).read()
    if "@@" in s:
        for k, val in v.items():
            s = s.replace(f"@@{k}@@", val)
        left = re.findall(r"@@\w+@@", s)
        assert not left, left
        open(p, "w").write(s)
        print("DESIGN.md filled;", len(v), "values")
    else:
        print("DESIGN.md has no placeholders and there is no docs/DESIGN.template.md: left as it is")
    # the round-6 block between the markers (text: docs/baseline_r6.md)
    bp = os.path.join(ROOT, "BASELINE.md")
    b = open(bp).read()
    blk = open(os.path.join(ROOT, "docs", "baseline_r6.md")).read()
    for k, val in v.items():
        blk = blk.replace(f"@@{k}@@", val)
    assert not re.findall(r"@@\w+@@", blk)
    m0, m1 = "<!-- R6 BEGIN -->\n", "<!-- R6 END -->\n"
    if m0 in b:
        b = b[:b.index(m0)] + m0 + blk + m1 + b[b.index(m1) + len(m1):]
    else:
        b = b.replace("## 3. Results\n\n", "## 3. Results\n\n" + m0 + blk + m1, 1)
    open(bp, "w").write(b)
    print("BASELINE.md round-6 block written")
    if "--table" in sys.argv:
        print(v["TABLE"])
