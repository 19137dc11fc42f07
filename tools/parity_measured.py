#!/usr/bin/env python
"""Aggregate the per-quantity errors a test run printed with PBRE_PARITY_MEASURE=1 (tests/parity.py: assert_within prints instead of
asserting) into one JSON: worst value per quantity and check.  The bounds in tests/parity.py (TOL_ICUB_CONTACT, TOL_ICUB_RESET,
TOL_HANDS_CONTACT, TOL_HANDS_RESET, ...) are ~4x the worst of the GPU run and the CPU lane emulation.
    PBRE_PARITY_MEASURE=1 python -m pytest tests/test_gpu_icub.py tests/test_gpu_hands.py tests/test_gpu_parity.py -m gpu -q -s > log
    python tools/parity_measured.py log > profiles/r03_parity_measured_hip.json"""
import ast, collections, json, re, sys
txt = open(sys.argv[1]).read()
groups = collections.OrderedDict()
for m in re.finditer(r"MEASURED (.*?): (\{.*?\})", txt):
    key = re.sub(r"\d+", "N", m.group(1))[:90]
    d = ast.literal_eval(m.group(2))
    g = groups.setdefault(key, {})
    for k, v in d.items():
        g[k] = max(g.get(k, 0.0), v)
extra = {}
for name, pat in (("panda_closed_loop_push", r"^\.*closed-loop push: (\{.*\})"), ("icub_closed_loop_push", r"iCub closed-loop push: (\{.*\})")):
    mm = re.findall(pat, txt, re.M)
    if mm:
        extra[name] = ast.literal_eval(mm[-1])
print(json.dumps({"source": sys.argv[1], "worst_per_check": groups, "closed_loop": extra}, indent=1))
