#!/usr/bin/env python3
"""gpurun_out/parity_true_shape.json (written by tests/test_gpu_model_parity.py on the GPU box) -> a markdown table
of the true-shape parity checks: error of the HIP path and of the reference's own fp32 golden against the fp64
oracle (max-norm relative), and the bound the test applied.
usage: python tools/parity_report.py gpurun_out/parity_true_shape.json profiles/r02_parity_true_shape.md"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
j = json.load(open(src))
lines = ["# True-shape parity (n = 400 dense-FC and n = 1000 kNN, 5 layers): HIP path vs fp64 oracle vs the reference's own fp32 golden", "",
         "Written by `tests/test_gpu_model_parity.py` on an MI355X; errors are max|a - ref| / max|ref| (gradients: with the",
         "2 % floor of tests/helpers.py `grad_floor`).  `reference` = the committed golden made by the real reference model in",
         "fp32; both columns are measured against `oracle/gin_oracle.py` in float64 on the same inputs.", ""]
for case in sorted(j):
    ent = j[case]
    lines += ["## %s" % case, "", "| tensor | HIP vs fp64 | HIP vs reference golden | reference vs fp64 | bound applied (vs fp64) |",
              "|---|---|---|---|---|"]
    for c in ent.get("checks", []):
        lines.append("| %s | %.2e | %s | %s | %.1e |" % (c["what"], c["hip_vs_fp64"],
                                                         "%.2e" % c["hip_vs_golden"] if c.get("hip_vs_golden") is not None else "-",
                                                         "%.2e" % c["reference_vs_fp64"] if c.get("reference_vs_fp64") is not None else "-",
                                                         c["bound"]))
    wg = ent.get("worst_gradient")
    if wg:
        lines += ["", "Worst parameter gradient of this case: `%s`" % json.dumps(wg)]
    lines.append("")
worst = max((c["hip_vs_fp64"] for e in j.values() for c in e.get("checks", [])), default=0.0)
worst_ref = max((c["reference_vs_fp64"] or 0.0 for e in j.values() for c in e.get("checks", [])), default=0.0)
lines += ["Largest activation / logit error over all cases: HIP %.2e, reference %.2e." % (worst, worst_ref), ""]
open(dst, "w").write("\n".join(lines))
print("\n".join(lines[-3:]))
