"""Random regular expressions (CPU only): patterns drawn from a grammar over literals, classes, escapes, POSIX classes, groups,
quantifiers and the assertions ^ $ \\A \\z \\b \\B; RE2 itself (pyarrow.compute.match_substring_regex: the RE2 in this image's libarrow) is
the reference; the planner-compiled position automaton (device function built for the host) and the oracle's thread list must
both agree with it on every text.      python tools/regex_fuzz.py <seed> <patterns>"""
import os, sys, ctypes as C, pyarrow as pa, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import test_registry_tail as T
from gandiva_amd import _capi
import gandiva_amd as g
from oracle import oracle
import pyarrow.compute as pc
lib=_capi.lib(); hostlib=C.CDLL('/root/repo/tests/host_devlib/libhost_devlib.so')
seed=int(sys.argv[1]); N=int(sys.argv[2])
rng=np.random.default_rng(seed)
atoms=["a","b","c",".","\\d","\\w","\\s","\\S","\\W","\\D","[ab]","[^a]","[a-c]","é","語","x","\\.","(ab|c)","(a|b)","(?:bc)","\\b","\\B","(^|-)","($|b)","^","$","\\A","\\z","[[:alpha:]]","[é語]","(?P<n>a|\\d)","-"," ","(é|\\b)","[^\\d]"]
quants=["","","","*","+","?","{2}","{1,3}","{2,}","*?","+?"]
texts=T._regex_texts(500,seed=seed)+T.REGEX_WORDS+["abcabc","aab","ccc","a.c","bcbc","é.é","a-b","ab ab","-a","b-","a\nb","語é語","é語a","aé","éa","\x0b","a\x0bb","\u212a","\u017f"]
arr=pa.array(texts,pa.string()); off=np.frombuffer(arr.buffers()[1],np.int32)[:len(texts)+1].copy(); size=int(off[-1])
data=np.concatenate([np.frombuffer(arr.buffers()[2],np.uint8)[:size],np.zeros(64,np.uint8)])
batch=pa.RecordBatch.from_arrays([arr],names=["s"]); b=g.TreeExprBuilder(); s=b.make_field(batch.schema.field(0))
p=lambda a:a.ctypes.data_as(C.c_void_p)
tried=bad=refused=0
for it in range(N):
    k=int(rng.integers(1,6)); pick=[atoms[int(rng.integers(0,len(atoms)))] for _ in range(k)]
    pat="".join(a+("" if a in ("\\b","\\B","^","$","\\A","\\z") else quants[int(rng.integers(0,len(quants)))]) for a in pick)
    if rng.random()<0.2: pat="(?"+("i" if "é" not in pat and "語" not in pat else "s")+")"+pat
    if rng.random()<0.15: pat="("+pat+")|zz"
    try: want=pc.match_substring_regex(arr,pat).to_pylist()
    except Exception as e: continue
    raw=pat.encode(); table=np.zeros(6360,np.uint8)
    if lib.gdv_compile_regex(raw,C.c_int64(len(raw)),p(table))!=0: refused+=1; continue
    tried+=1
    out=np.zeros(len(texts),np.uint8)
    hostlib.host_regex_search(p(off),p(data),C.c_long(size),C.c_long(len(texts)),p(table),0,p(out))
    orc=oracle.project([T._regex_expr(b,s,pat)],batch)[0].to_pylist()
    if out.astype(bool).tolist()!=want or orc!=want:
        bad+=1
        if bad<8:
            i=[j for j in range(len(texts)) if bool(out[j])!=want[j] or orc[j]!=want[j]][0]
            print("MISMATCH",repr(pat),repr(texts[i]),"dev",bool(out[i]),"oracle",orc[i],"re2",want[i])
print("seed",seed,"tried",tried,"refused by the compiler",refused,"mismatching",bad)
