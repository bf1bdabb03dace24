"""A longer differential-fuzz campaign than tests/test_gpu_fuzz.py runs by default (manual tool):

    python tests/fuzz_campaign.py [first_seed=100] [items=60]        # on a GPU box

Random work items of random shape (dtype, inputs, 60-3500 nodes, outputs, setters, 1-5001 rays),
separate and fused launches, bit for bit against the CPU oracle.
"""
import sys, time
import os
HERE=os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0,os.path.dirname(HERE)); sys.path.insert(0,HERE)
import numpy as np
import gfir_random
from oracle import gfir
from graph_framework_amd import Context
fails=0; total=0; one_piece=0
t0=time.time()
first=int(sys.argv[1]) if len(sys.argv)>1 else 100
count=int(sys.argv[2]) if len(sys.argv)>2 else 60
for seed in range(first, first+count):
    rng=np.random.default_rng(seed)
    dtype='f64' if rng.random()<0.7 else 'f32'
    inputs=int(rng.integers(2,9)); nodes=int(rng.choice([60,150,400,900,2000,3500]))
    outputs=int(rng.integers(1,5)); setters=int(rng.integers(0,inputs+1)); rays=int(rng.choice([1,63,64,65,257,1000,5001]))
    blob,_=gfir_random.random_item(seed,dtype,inputs,nodes,outputs,setters)
    item=gfir.Item(blob)
    init=[rng.uniform(-1,1,rays).astype(item.np_dtype) for _ in range(inputs)]
    ctx=Context(0); k=ctx.add_kernel(blob,rays); ctx.compile()
    ink=['i%d'%i for i in range(inputs)]; outk=['o%d'%i for i in range(outputs)]
    k.create_kernel_call(ink,outk,init)
    one_piece+=k.info().segments==1          # one piece + redo launch: the assembly body (csrc/asm_body.hpp) took the item
    exp=[c.copy() for c in init]; ok=True
    for steps in (1,2):
        eo,_=item.run(exp,steps=steps); k.run(steps); ctx.wait()
        if ctx.flags()&1: print('seed',seed,'took the IEEE function (status bit 0: window/finite)')     # informational: the values decide
        for key,want in zip(ink+outk, exp+eo):
            got=ctx.copy_to_host(key,np.empty(rays,dtype=item.np_dtype))
            if not np.array_equal(got,want): ok=False
    ctx.close(); total+=1
    if not ok: fails+=1; print('MISMATCH seed',seed,dtype,inputs,nodes,outputs,setters,rays, flush=True)
    if total%10==0: print(total,'items',time.time()-t0,'s',flush=True)
print('campaign: %d items (%d with the assembly body), %d mismatches'%(total,one_piece,fails))
sys.exit(1 if fails else 0)
