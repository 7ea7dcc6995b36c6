"""Generate the reference's C++ header nlopt.hpp from the sources where they lie (TEST INFRASTRUCTURE, build container only).
The reference produces it with cmake/generate-cpp.cmake: src/api/nlopt-in.hpp is copied and, at the GEN_ENUMS_HERE marker,
the `algorithm` and `result` enums are emitted from the NLOPT_* enumerators of src/api/nlopt.h with the prefix removed.
We do not run cmake (oracle/Makefile); this script does that one transformation.
usage: gen_nlopt_hpp.py <reference root> <output nlopt.hpp>"""
import re
import sys

ref, out = sys.argv[1], sys.argv[2]
hpp = open(ref + "/src/api/nlopt-in.hpp").read().split("\n")
h = open(ref + "/src/api/nlopt.h").read().split("\n")
res = []
for line in hpp:
    res.append(line)
    if "GEN_ENUMS_HERE" in line:
        res.append("  enum algorithm {")
        for hl in h:
            if re.search(r"^    NLOPT_[A-Z0-9_]+", hl):
                res.append(hl.replace("NLOPT_", "", 1))
                if "NLOPT_NUM_ALGORITHMS" in hl:
                    res += ["  };", "  enum result {"]
                elif "NLOPT_NUM_RESULTS" in hl:
                    res.append("  };")
open(out, "w").write("\n".join(res))
