"""Mutation fuzzing of the GFIR parser and lowering (run by tests/test_cabi.py in a child process so
that a crash cannot take the test runner down): truncations, byte flips and corrupted 32-bit fields
of a valid item must be either lowered or rejected with an error, never crash.

    python tests/gfir_mutate.py <seed> <trials>
"""
import ctypes
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import gfir_random
from graph_framework_amd import _lib
lib=_lib.load()
lib.gfhip_generate_source.restype=ctypes.c_void_p
lib.gfhip_generate_source.argtypes=[ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
lib.gfhip_free_string.argtypes=[ctypes.c_void_p]
rnd=random.Random(int(sys.argv[1]))
blob,_=gfir_random.random_item(3,'f64',num_nodes=200)
base=bytearray(blob)
ok=bad=0
for trial in range(int(sys.argv[2])):
    b=bytearray(base)
    kind=rnd.random()
    if kind<0.3:
        b=b[:rnd.randrange(0,len(b))]
    elif kind<0.8:
        for _ in range(rnd.randrange(1,6)):
            pos=rnd.randrange(0,len(b)); b[pos]=rnd.randrange(256)
    else:
        # corrupt a 32-bit field with a large value
        pos=rnd.randrange(0,len(b)//4)*4; b[pos:pos+4]=(rnd.choice([0xFFFFFFFF,0x7FFFFFFF,0x80000000,100000,rnd.randrange(2**32)])).to_bytes(4,'little')
    h=ctypes.c_uint64()
    p=lib.gfhip_generate_source(bytes(b),len(b),ctypes.byref(h))
    if p: ok+=1; lib.gfhip_free_string(p)
    else: bad+=1
print('accepted',ok,'rejected',bad)
