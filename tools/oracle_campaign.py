"""The oracle's cross-checks over seeds the suite does not use (CPU only): random numeric trees vs pyarrow.compute, random string trees and
trees with materialised values vs plain Python (the generators of tests/test_oracle_vs_arrow_trees.py and tests/test_oracle_vs_python_strings.py).
      python tools/oracle_campaign.py <first_seed> <last_seed>"""
import sys, traceback
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import pytest
import test_oracle_vs_arrow_trees as A
import test_oracle_vs_python_strings as P
import test_fuzz_trees as F
lo,hi=int(sys.argv[1]),int(sys.argv[2])
# apply the fixture's restriction by hand
F.EXACT = A.ARROW_OPS
bad=0
for seed in range(lo,hi):
    for name,fn in (("arrow",A.test_oracle_trees_match_pyarrow_compute),("strings",P.test_oracle_string_trees_match_plain_python),("materialised",P.test_oracle_trees_with_materialised_values_match_plain_python)):
        try: fn(seed)
        except Exception as e:
            bad+=1; print("FAIL",name,seed,type(e).__name__,str(e)[:300].replace("\n"," "),flush=True)
print("seeds",lo,hi,"failures",bad)
