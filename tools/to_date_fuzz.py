"""Random to_date patterns (CPU only): token sequences with separators and quoted text, texts rendered by strftime and mutated; the
device library's interpreter (host build of gdv_parse_date over gdv_compile_date_format's program) against the oracle, which calls
glibc's strptime — the function the lineage's holder reaches through Arrow.      python tools/to_date_fuzz.py <seed> <patterns>"""
import sys, ctypes as C, numpy as np, pyarrow as pa, datetime
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import test_registry_tail as T
import gandiva_amd as g
from gandiva_amd import _capi
from oracle import oracle
hostlib=C.CDLL('/root/repo/tests/host_devlib/libhost_devlib.so')
rng=np.random.default_rng(int(sys.argv[1]))
tokens=[("YYYY","%Y"),("YY","%y"),("MM","%m"),("MON","%b"),("MONTH","%B"),("DD","%d"),("DDD","%j"),("DY","%a"),("DAY","%A"),("HH24","%H"),("HH","%I"),("HH12","%I"),("MI","%M"),("SS","%S"),("AM","%p"),("PM","%p")]
seps=["-","/"," ",":",", ",".","T"," - ",""]
p=lambda a:a.ctypes.data_as(C.c_void_p)
bad=0; tried=0
for it in range(int(sys.argv[2])):
    k=int(rng.integers(1,6)); parts=[]; pyf=[]
    for j in range(k):
        sql,py=tokens[int(rng.integers(0,len(tokens)))]
        sep=seps[int(rng.integers(0,len(seps)))] if j<k-1 else ""
        if sep=="T": parts.append(sql+'"T"'); pyf.append(py+"T")
        else: parts.append(sql+sep); pyf.append(py+sep)
    pattern="".join(parts); pyfmt="".join(pyf)
    texts=[]
    for _ in range(120):
        t=datetime.datetime(1970,1,1)+datetime.timedelta(days=int(rng.integers(-20000,30000)),seconds=int(rng.integers(0,86400)))
        s=t.strftime(pyfmt)
        r=rng.random()
        if r<0.3:
            pos=int(rng.integers(0,len(s)+1)); kind=int(rng.integers(0,4))
            s = s[:pos] if kind==0 else s[:pos]+"xX9 -:/"[int(rng.integers(0,7))]+s[pos+1:] if kind==1 else s[:pos]+" "+s[pos:] if kind==2 else s.upper()
        texts.append(s)
    arr=pa.array(texts,pa.string()); batch=pa.RecordBatch.from_arrays([arr],names=["s"])
    b=g.TreeExprBuilder()
    try: want=oracle.project([T._to_date_exprs(b,b.make_field(batch.schema.field(0)),pattern,1)],batch)[0].cast(pa.int64()).to_pylist()
    except Exception as e: print("oracle refused",repr(pattern),e); continue
    raw=pattern.encode(); buf=np.zeros(256,np.uint8); cnt=C.c_int64(0)
    if _capi.lib().gdv_compile_date_format(raw,len(raw),p(buf),248,C.byref(cnt))!=0: print("planner refused",repr(pattern)); continue
    off=np.frombuffer(arr.buffers()[1],np.int32)[:len(texts)+1].copy(); size=int(off[-1])
    data=np.concatenate([np.frombuffer(arr.buffers()[2],np.uint8)[:size],np.zeros(64,np.uint8)])
    out,ov=np.zeros(len(texts),np.int64),np.zeros(len(texts),np.uint8)
    hostlib.host_parse_date(p(off),p(data),C.c_long(size),C.c_long(len(texts)),p(buf),C.c_int(cnt.value),1,p(out),p(ov))
    got=[int(v) if ok else None for v,ok in zip(out,ov)]
    tried+=1
    if got!=want:
        bad+=1
        if bad<8:
            i=[j for j in range(len(texts)) if got[j]!=want[j]][0]; print("MISMATCH",repr(pattern),repr(texts[i]),"dev",got[i],"glibc",want[i])
print("patterns",tried,"mismatching",bad)
