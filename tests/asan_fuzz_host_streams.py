"""Run by tests/test_sanitizer_cpu.py under LD_PRELOAD=libasan: the product's host-side decoders (leon_amd/csrc/host_streams.cpp,
built with -fsanitize=address,undefined) on valid, bit-flipped, truncated and random payloads -- any answer but a memory error."""
import ctypes as C, os, sys, zlib, random
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT)
import oracle_lib as O, hdr_samples as H
L=C.CDLL(sys.argv[1])
u8p=C.POINTER(C.c_uint8); u64p=C.POINTER(C.c_uint64); u32p=C.POINTER(C.c_uint32)
L.leon_host_header_decode_blocks.argtypes=[u8p,u64p,u32p,C.c_uint64,C.c_char_p,C.c_uint64,u8p,C.c_uint64,u64p,u64p,C.c_uint32]
def dec(blocks, first, threads=2):
    pay=np.frombuffer(b"".join(b[1] for b in blocks)+b"\0",dtype=np.uint8)
    off=np.zeros(len(blocks)+1,dtype=np.uint64); off[1:]=np.cumsum([len(b[1]) for b in blocks])
    nr=np.array([b[2] for b in blocks],dtype=np.uint32); total=int(nr.sum())
    out_off=np.zeros(total+1,dtype=np.uint64); need=C.c_uint64(); cap=64*total+64
    for _ in range(2):
        out=np.zeros(cap,dtype=np.uint8)
        rc=L.leon_host_header_decode_blocks(pay.ctypes.data_as(u8p),off.ctypes.data_as(u64p),nr.ctypes.data_as(u32p),len(blocks),first,len(first),out.ctypes.data_as(u8p),cap,out_off.ctypes.data_as(u64p),C.byref(need),threads)
        if rc!=-5: break
        cap=need.value
    if rc: return rc
    raw=out.tobytes(); return [raw[int(out_off[i]):int(out_off[i+1])] for i in range(total)]
for make,n,rpb in ((H.sra,3000,700),(H.nasty,900,64),(H.toy_like,50,50000)):
    hs=make(n); blocks=[(b//rpb,O.header_encode_block(hs[b:b+rpb],hs[0]),len(hs[b:b+rpb])) for b in range(0,n,rpb)]
    assert dec(blocks,hs[0])==hs
    rnd=random.Random(1)
    for trial in range(300):   # corrupted / truncated payloads: any answer but a memory error
        bb=list(blocks); i=rnd.randrange(len(bb)); p=bytearray(bb[i][1])
        kind=rnd.randrange(3)
        if kind==0 and p: p[rnd.randrange(len(p))]^=1<<rnd.randrange(8)
        elif kind==1: p=p[:rnd.randrange(len(p)+1)]
        else: p=bytearray(rnd.randrange(256) for _ in range(rnd.randrange(1,200)))
        bb[i]=(bb[i][0],bytes(p),bb[i][2]); dec(bb,hs[0])
print("header decoder under ASan/UBSan: ok")
L.leon_host_qual_decode_blocks.argtypes=[u8p,u64p,u32p,u64p,C.c_uint64,u8p,C.c_uint64,u64p,C.c_uint32]
qs=H.fastq_quals(500,80); rpb=100
blocks=[(b//rpb, zlib.compress(b"".join(q+b"\n" for q in qs[b:b+rpb])), len(qs[b:b+rpb])) for b in range(0,500,rpb)]
def qdec(blocks,nbytes):
    pay=np.frombuffer(b"".join(b[1] for b in blocks)+b"\0",dtype=np.uint8)
    off=np.zeros(len(blocks)+1,dtype=np.uint64); off[1:]=np.cumsum([len(b[1]) for b in blocks])
    nr=np.array([b[2] for b in blocks],dtype=np.uint32); nb=np.array(nbytes,dtype=np.uint64)
    out=np.zeros(int(nb.sum())+1,dtype=np.uint8); oo=np.zeros(int(nr.sum())+1,dtype=np.uint64)
    return L.leon_host_qual_decode_blocks(pay.ctypes.data_as(u8p),off.ctypes.data_as(u64p),nr.ctypes.data_as(u32p),nb.ctypes.data_as(u64p),len(blocks),out.ctypes.data_as(u8p),int(nb.sum()),oo.ctypes.data_as(u64p),2)
nbytes=[sum(len(q) for q in qs[b:b+rpb]) for b in range(0,500,rpb)]
assert qdec(blocks,nbytes)==0
rnd=random.Random(2)
for trial in range(200):
    bb=list(blocks); i=rnd.randrange(len(bb)); text=bytearray(zlib.decompress(bb[i][1]))
    k=rnd.randrange(4)
    if k==0: text=text.replace(b"\n",b"",1)
    elif k==1: text+=b"\n\n"
    elif k==2: text=text[:rnd.randrange(len(text))]
    else: text[rnd.randrange(len(text))]=10
    bb[i]=(bb[i][0],zlib.compress(bytes(text)),bb[i][2]); qdec(bb,nbytes)
    nb2=list(nbytes); nb2[rnd.randrange(5)]=rnd.randrange(20000); qdec(blocks,nb2)
print("quality decoder under ASan/UBSan: ok")
