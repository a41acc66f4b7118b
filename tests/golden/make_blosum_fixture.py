"""Writes tests/golden/blosum_tables.json from the reference's tables (run in the build container, where /root/reference exists):
the numbers of BLOSUM45 / 62 / 80 in ACDEFGHIKLMNPQRSTVWY order exactly as /root/reference/src/blosum.hpp:9-79 holds them
(BLOSUM80 there is not symmetric: [V][I] = 1, [I][V] = 3).  The host mirror's msa::Params must reproduce 5 x these."""
import json
import os
import re

src = open("/root/reference/src/blosum.hpp").read()
out = {}
for name in ("BLOSUM45", "BLOSUM62", "BLOSUM80"):
    body = re.search(r"const float " + name + r"\[20\]\[20\] = \{(.*?)\};", src, re.S).group(1)
    nums = [int(x) for x in re.findall(r"-?\d+", re.sub(r"//.*", "", body))]
    assert len(nums) == 400
    out[name[6:]] = [nums[i * 20:(i + 1) * 20] for i in range(20)]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "blosum_tables.json"), "w"))
