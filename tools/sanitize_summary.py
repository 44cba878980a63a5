#!/usr/bin/env python3
"""Summary of a sanitizer log (tools/sanitize.sh run ...).  A report has one stack per access; the frame that matters is
the first one below the sanitizer's own interceptors: the code that performed the racing (or invalid) access.  Reports
are counted by the module of that frame -- the library / the driver ("ours") or the uninstrumented HIP / HSA runtime,
whose internal accesses ThreadSanitizer only sees through intercepted libc calls (memcpy, free, pthread_*), with this
library merely further up the stack as the caller of a HIP API."""
import re
import sys

txt = open(sys.argv[1], errors="replace").read()
reports = [r for r in re.split(r"={18}\n", txt) if "WARNING:" in r or "ERROR:" in r]


def owner(frame):
    if "libwspr_mi355x" in frame or "sanitize_driver" in frame or "(driver+" in frame:
        return "ours"
    m = re.search(r"\(([^()\s]+?\.so[0-9.]*)\+0x", frame)
    return m.group(1) if m else "other"


by_kind, ours = {}, []
for r in reports:
    kind = re.search(r"(?:WARNING|ERROR): (\w+Sanitizer: [^\n(]*)", r)
    kind = kind.group(1).strip() if kind else "?"
    stacks, cur = [], None
    for ln in r.splitlines():
        if re.match(r"\s+#\d+ ", ln):
            if cur is not None:
                cur.append(ln.strip())
        elif ln.startswith("  ") and ln.rstrip().endswith(":") and ("thread" in ln or "of size" in ln or "by" in ln):
            cur = []
            stacks.append((ln.strip(), cur))
    access = [s for s in stacks if re.search(r"of size|Atomic|[Rr]ead|[Ww]rite|free", s[0])][:2]
    owners = []
    for title, fr in access:
        first = next((f for f in fr if "libclang_rt" not in f), fr[0] if fr else "")
        owners.append(owner(first))
    key = (kind, tuple(sorted(set(owners))) or ("?",))
    by_kind[key] = by_kind.get(key, 0) + 1
    if "ours" in owners:
        ours.append((kind, [f for _, fr in access for f in fr[:3]]))
print("%d reports; by kind and by the module of the code that made the accesses:" % len(reports))
for (kind, own), n in sorted(by_kind.items(), key=lambda kv: -kv[1]):
    print("  %4d  %-44s accesses in: %s" % (n, kind, ", ".join(own)))
both_ours = sum(n for (kind, own), n in by_kind.items() if set(own) == {"ours"})
print("%d report(s) with an access made by the library's or the driver's own code; %d with BOTH accesses there (a race of "
      "this code with itself)" % (len(ours), both_ours))
for kind, fr in ours[:10]:
    print("   " + kind)
    for f in fr:
        print("        " + f[:220])
print("\n".join(ln for ln in txt.splitlines() if "SANITIZE DRIVER" in ln or "decode their message" in ln or "hashed calls resolved" in ln))
