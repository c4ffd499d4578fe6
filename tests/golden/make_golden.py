"""
tests/golden/make_golden.py -- regenerates tests/golden/reference_notebook.json.

The reference (a Rust crate) cannot be built or imported in this image, so the only literal
reference OUTPUTS available are the ones its authors committed: the rendered cells of
/root/reference/examples/basics.ipynb (polars prints 6 significant digits).  This script parses the
printed tables of the lin_reg cells and stores inputs and outputs verbatim.  Run it in the build
container (it reads /root/reference, which does not exist on the GPU box); the JSON is committed.
"""
import json
import re
import sys
from pathlib import Path

NB = Path("/root/reference/examples/basics.ipynb")
OUT = Path(__file__).resolve().parent / "reference_notebook.json"


def cell_outputs(nb):
    for c in nb["cells"]:
        if c["cell_type"] != "code":
            continue
        src = "".join(c["source"])
        txt = ""
        for o in c.get("outputs", []):
            t = o.get("text") or o.get("data", {}).get("text/plain")
            if t:
                txt += "".join(t)
        yield src, txt


def rows(txt):
    out = []
    for line in txt.splitlines():
        if line.startswith("│") and not any(k in line for k in ("---", "f64", "str", "list[")):
            out.append([f.strip() for f in line.strip("│").split("┆")])
    return out


def main():
    nb = json.loads(NB.read_text())
    gold = {"source": "examples/basics.ipynb (reference v0.12.1), printed cell outputs, 6 significant digits"}
    for src, txt in cell_outputs(nb):
        if "return_pred=True" in src and ".head()" in src and "lin_reg(\"x1\"" in src.replace("'", '"'):
            r = [x for x in rows(txt) if len(x) == 5 and x[0] not in ("x1",)]
            a = [[float(v) for v in x] for x in r]
            gold["pred_head"] = {"x1": [x[0] for x in a], "x2": [x[1] for x in a], "y": [x[2] for x in a],
                                 "pred": [x[3] for x in a], "resid": [x[4] for x in a]}
        if src.lstrip().startswith("# Linear Regression\n"):
            m = re.search(r"\[(-?[\d.]+), (-?[\d.]+)\]", txt)
            gold["lin_reg_full_frame_coeffs"] = [float(m.group(1)), float(m.group(2))]
        if "lin_reg_report" in src and "ln(x1+1)" in src:
            # the printed report table (f64 cell, then the f32 cell): features | beta | std_err | t | ... | 0.025 | 0.975 | r2 | adj_r2
            # (polars elides the middle columns with an ellipsis); 10 000 rows, 3 features + bias -> dof = 9 996
            r = [x for x in rows(txt) if len(x) == 9 and x[0] not in ("features",)]
            key = "report_f32" if "LIN_REG_EXPR_F64 = False" in src else "report_f64"
            gold[key] = [{"feature": x[0], "beta": float(x[1]), "std_err": float(x[2]), "t": float(x[3]), "ci_lo": float(x[5]),
                          "ci_hi": float(x[6]), "r2": float(x[7]), "adj_r2": float(x[8])} for x in r]
            gold["report_rows"], gold["report_dof"] = 10_000, 9_996
        if "target=[pl.col(\"y\"), pl.col(\"y2\")]" in src and "LIN_REG_EXPR_F64" not in src:
            m = re.findall(r"\[(-?[\d.]+), (-?[\d.]+)\]", txt)
            gold["multi_target_coeffs"] = [[float(a), float(b)] for a, b in m]
        if "rolling_lin_reg" in src and "window_size=5" in src:
            r = [x for x in rows(txt) if len(x) == 5 and x[0] not in ("y", "…")]
            parsed = []
            for x in r:
                co = None if x[3] == "null" else [float(v) for v in x[3].strip("[]").split(",")]
                parsed.append({"y": float(x[0]), "x1": float(x[1]), "x2": float(x[2]), "coeffs": co,
                               "pred": None if x[4] == "null" else float(x[4])})
            gold["rolling_w5_head"] = parsed[:5]
            gold["rolling_w5_tail"] = parsed[5:]
    OUT.write_text(json.dumps(gold, indent=1))
    print("wrote", OUT, list(gold))


if __name__ == "__main__":
    sys.exit(main())
