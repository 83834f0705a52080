#!/usr/bin/env python3
"""dev aid (no GPU): gzip files of several members through `barbell-amd stage`, the members inflated side by side (ParallelInflater::
inflate_regular: ranges of the compressed bytes, a range's first member guessed by the gzip magic and accepted only if the chain links up) against one
after the other (BARBELL_AMD_GZ_SERIAL=1) and against the text itself: members cut at arbitrary points of the text, stored members whose text
holds the magic bytes (false starts), ranges of 64 bytes up (BARBELL_AMD_GZ_RANGE), pieces of 64 bytes up, trailing garbage (ignored, as gzread
does), truncated files (an error either way).  usage: gz_members_fuzz.py FIRST_SEED N_CASES"""
import gzip, os, subprocess, sys, zlib, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, 'barbell_amd', 'bin', 'barbell-amd')
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
def rec(i, L, evil=False):
    s=bytes(rng.choice(np.frombuffer(b"ACGT",dtype=np.uint8),L))
    h=b"@r%d"%i + (b" \x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03junk" if evil else b"")
    return h+b"\n"+s+b"\n+\n"+b"I"*L+b"\n"
def stage(path, env):
    r=subprocess.run([CLI,'stage','-i',path,'-o','/tmp/gzpar.out','--block-bytes','65536','-t','6','--no-compact'],capture_output=True,text=True,env=dict(os.environ,BARBELL_AMD_PROFILE='1',**env),timeout=120)
    m=re.search(r"(\d+) range\(s\) of members",r.stderr)
    return r.returncode, open('/tmp/gzpar.out','rb').read() if r.returncode==0 else r.stderr[-300:], int(m.group(1)) if m else -1
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 40):
    nm=int(rng.integers(1,40))
    members=[]; text=b""
    for m in range(nm):
        t=b"".join(rec(m*1000+i,int(rng.integers(1,600)),evil=rng.random()<0.2) for i in range(int(rng.integers(0,30))))
        # members need not end on record boundaries: cut the text stream at arbitrary points instead
        text+=t
    cuts=sorted(set([0,len(text)]+[int(x) for x in rng.integers(0,len(text)+1,nm-1)])) if len(text) else [0,0]
    blob=b""
    for a,b in zip(cuts[:-1],cuts[1:]):
        lvl=int(rng.choice([0,1,6]))
        blob+=gzip.compress(text[a:b],compresslevel=lvl)
    kind=rng.choice(["ok","garbage","trunc"],p=[0.7,0.15,0.15])
    if kind=="garbage": blob+=b"\x00"*int(rng.integers(1,2000))
    if kind=="trunc" and len(blob)>40: blob=blob[:len(blob)-int(rng.integers(1,min(len(blob)-20,300)))]
    open('/tmp/gzpar.gz','wb').write(blob)
    RB=str(int(rng.choice([64,200,1000,5000])))
    rc_s,out_s,_=stage('/tmp/gzpar.gz',{'BARBELL_AMD_GZ_SERIAL':'1','BARBELL_AMD_GZ_PIECE':'3000'})
    rc_p,out_p,nr=stage('/tmp/gzpar.gz',{'BARBELL_AMD_GZ_RANGE':RB,'BARBELL_AMD_GZ_PIECE':str(int(rng.choice([64,3000,100000])))})
    if kind!="trunc":
        okk = rc_s==0 and rc_p==0 and out_s==text and out_p==text
    else:
        okk = (rc_s!=0)==(rc_p!=0) and (rc_s!=0 or out_s==out_p)
    if not okk:
        bad+=1; print("BAD",it,kind,nm,RB,rc_s,rc_p,len(text), len(out_s) if rc_s==0 else out_s, len(out_p) if rc_p==0 else out_p)
        os.system(f"cp /tmp/gzpar.gz /tmp/gzpar_bad{it}.gz")
    elif it<8: print(it,kind,"members",len(cuts)-1,"bytes",len(blob),"RB",RB,"ranges in parallel",nr)
print("bad",bad)
