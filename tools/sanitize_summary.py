#!/usr/bin/env python3
"""Summary of a sanitizer log (tools/sanitize.sh run ...): reports by SUMMARY line, and -- what matters -- the reports
with a frame of THIS library's own code in either stack (libwspr_mi355x_lab.so / wspr:: / sanitize_driver), apart from
the HIP module constructor / destructor the runtime registers for every shared object."""
import re
import sys

txt = open(sys.argv[1], errors="replace").read()
reports = re.split(r"={18}\n", txt)
reports = [r for r in reports if "WARNING:" in r or "ERROR:" in r]
by_summary, ours = {}, []
for r in reports:
    m = re.search(r"SUMMARY: (\w+Sanitizer: [^\n(]*)\(([^)+]*)", r)
    key = (m.group(1).strip() + " in " + m.group(2).split("/")[-1]) if m else "no summary"
    by_summary[key] = by_summary.get(key, 0) + 1
    frames = [ln for ln in r.splitlines() if re.match(r"\s+#\d+", ln)]
    mine = [f for f in frames if ("libwspr_mi355x" in f or "wspr::" in f or "sanitize_driver" in f)
            and "__hip_module_ctor" not in f and "__hip_module_dtor" not in f]
    if mine:
        ours.append((key, mine[:4]))
print("%d reports" % len(reports))
for k, v in sorted(by_summary.items(), key=lambda kv: -kv[1]):
    print("  %4d  %s" % (v, k))
print("%d of them with a frame of the library's or the driver's own code (module constructors excluded):" % len(ours))
seen = {}
for k, fr in ours:
    sig = (k, tuple(re.sub(r"0x[0-9a-f]+", "", f).strip() for f in fr[:2]))
    seen[sig] = seen.get(sig, 0) + 1
for (k, fr), n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("  %3d x %s" % (n, k))
    for f in fr:
        print("        " + f[:200])
last = [ln for ln in txt.splitlines() if "SANITIZE DRIVER" in ln or "decode their message" in ln or "hashed calls resolved" in ln]
print("\n".join(last))
