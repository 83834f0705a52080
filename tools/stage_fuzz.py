#!/usr/bin/env python3
"""dev aid (no GPU): randomised check of the C++ host's packed staging (`barbell-amd stage`, host/bb_feed.cpp compact_two_line + PackCtx) against the
Python packer (barbell_amd/fastq.py): random record lengths around the 32-base vector width, LF / CRLF, IUPAC and non-IUPAC characters, files
with and without a final newline, blank lines after the last record, gzip (inflated in pieces of 64 bytes up), several files, chunk sizes from 17 bytes up, 1-5 reader threads.
Every other seed also checks --shard R/W --shard-by bytes (the shards' texts concatenated = the unsharded text).
Round 5 found three chunk-boundary bugs with it (a "\r" | "\n" split over two chunks, blank tail lines straddling a chunk start).
usage: stage_fuzz.py FIRST_SEED N_SEEDS"""
import sys, os, subprocess, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from barbell_amd import fastq as Q
CLI = os.path.join(ROOT, 'barbell_amd', 'bin', 'barbell-amd')
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[1])+int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    nl=b"\r\n" if rng.random()<0.3 else b"\n"
    nfiles=int(rng.integers(1,4))
    files=[]; want=b""; ok=True
    for f in range(nfiles):
        n=int(rng.integers(0,60))
        recs=[]
        for i in range(n):
            L=int(rng.choice([0,1,2,3,31,32,33,63,64,65,int(rng.integers(0,400)),int(rng.integers(0,3000))]))
            alpha=rng.choice([b"ACGT",b"ACGTNacgtn",b"ACGTRYKMSWBDHVNU*-"],p=[0.6,0.3,0.1])
            s=rng.choice(np.frombuffer(bytes(alpha),dtype=np.uint8),L).tobytes()
            recs.append(((b"r%d_%d"%(f,i))+(b" x y" if i%3==0 else b""), s))
        text=b"".join(b"@"+h+nl+s+nl+b"+"+nl+b"I"*len(s)+nl for h,s in recs)
        final_nl = rng.random()<0.8
        if not final_nl and recs and len(recs[-1][1])>0: text=text[:-len(nl)]
        tail=b""
        if final_nl and recs:
            k=int(rng.choice([0,0,1,2,3,4,5,7]))      # blank lines after the last record: the two-line form keeps those in header / sequence phase
            text+=nl*k
            tail=nl*sum(1 for i in range(k) if i%4<2)
        p=f"/tmp/barbell_stage_fuzz/f{f}.fq"+(".gz" if rng.random()<0.25 else "")
        os.makedirs('/tmp/barbell_stage_fuzz',exist_ok=True)
        with (gzip.open(p,'wb') if p.endswith('.gz') else open(p,'wb')) as fh: fh.write(text)
        files.append(p)
        pk=Q.pack_two_line(recs,nl)
        if pk is None: ok=False
        else: want+=pk+tail
    block=int(rng.choice([17,64,100,257,1000,4096,1<<20]))
    t=int(rng.integers(1,6))
    env=dict(os.environ)
    if rng.random()<0.7: env['BARBELL_AMD_GZ_PIECE']=str(int(rng.choice([64,100,333,1000,5000,70000])))   # gzip input comes in record-aligned pieces (ParallelInflater): tiny ones
    r=subprocess.run([CLI,'stage','-i']+files+['-o','/tmp/barbell_stage_fuzz/o.bin','--block-bytes',str(block),'-t',str(t)],capture_output=True,text=True,env=env,timeout=120)   # (a hang is a failure: the traceback names the seed)
    if r.returncode!=0:
        print('seed',seed,'rc',r.returncode,r.stderr[-200:]); bad+=1; continue
    form=int(r.stdout.split()[1]); got=open('/tmp/barbell_stage_fuzz/o.bin','rb').read()
    if ok:
        if form!=1 or got!=want:
            print('seed',seed,'MISMATCH form',form,len(got),len(want),'block',block,'t',t,'nl',nl,files); bad+=1
    else:
        if form!=2: print('seed',seed,'expected fallback, form',form); bad+=1
    # --shard R/W --shard-by bytes (plain files): the shards' staged texts, one after the other, are the unsharded one
    if r.returncode==0 and not any(p.endswith('.gz') for p in files) and seed%2==0:
        W=int(rng.choice([2,3,7,40])); cat=b""; fine=True
        one=[]
        for p in files:      # per file, as the shards of a multi-file run interleave (file 0 shard r, file 1 shard r, ...)
            parts=[]
            for R in range(W):
                rr=subprocess.run([CLI,'stage','-i',p,'-o','/tmp/barbell_stage_fuzz/s.bin','--block-bytes',str(block),'-t',str(t),'--shard',f'{R}/{W}','--shard-by','bytes']+([] if form==1 else ['--no-pack']),capture_output=True,text=True)
                if rr.returncode!=0: print('seed',seed,'shard',R,W,'rc',rr.returncode,rr.stderr[-200:]); fine=False; break
                parts.append(open('/tmp/barbell_stage_fuzz/s.bin','rb').read())
            one.append(b"".join(parts))
        if fine:
            if b"".join(one)!=got: print('seed',seed,'SHARD MISMATCH',W,block,files); bad+=1
        else: bad+=1
    for p in files: os.remove(p)
print(int(sys.argv[2]),'seeds',bad,'bad')
