"""Host-only entry points of libdcsg_hip under AddressSanitizer + UBSan (CPU build; see tools/sanitize_host.sh):
every fixture scene and ~1300 mutated / truncated inputs through the parsers and the program optimiser, the
export-config and lookup-table parsers, all four writers."""
import ctypes as C, os, sys, numpy as np, glob
L=C.CDLL(os.environ.get("DCSG_ASAN_LIB", "/tmp/dcsg_asan/libdcsg_hip.so"))
L.dcsg_compile_program.argtypes=[C.c_char_p,C.c_char_p,C.c_void_p,C.POINTER(C.c_int),C.POINTER(C.c_int),C.c_void_p,C.POINTER(C.c_int),C.POINTER(C.c_int),C.c_int]
def compile(st,bp):
    reg=(C.c_int32*(4*8192))(); stk=(C.c_int32*(4*8192))(); a=C.c_int(); b=C.c_int(); c=C.c_int(); d=C.c_int()
    return L.dcsg_compile_program(st,bp,reg,C.byref(a),C.byref(b),stk,C.byref(c),C.byref(d),8192)
n=0
for d in sorted(glob.glob("/root/repo/tests/golden/scenes/*/")):
    st=open(d+"scene.txt","rb").read(); bp=open(d+"buildprocedure.txt","rb").read()
    assert compile(st,bp)==0, d; n+=1
    # mutations: truncated, garbage tokens, huge numbers, negative slots
    rng=np.random.default_rng(1)
    for k in range(60):
        b=bytearray(bp); 
        if len(b)>4:
            for _ in range(3):
                i=int(rng.integers(0,len(b))); b[i]=int(rng.choice([ord('9'),ord('-'),ord(' '),ord('x'),ord('\n'),ord('7')]))
        compile(st,bytes(b[:int(rng.integers(0,len(b)+1))])); compile(st[:int(rng.integers(0,len(st)+1))],bytes(b)); n+=2
# opcodes outside the file format (incl. the host's own fused / stack numbers) with wild operands: dropped, not indexed
st=open("/root/repo/tests/golden/scenes/mixed_groups/scene.txt","rb").read()
junk=["0 0 0 0"]+["%d 2 %d %d"%(op,5000+op,(-1)**op*(70000+op)) for op in list(range(9,64))+[-1,255,100000]]+["16 2 0 5000","0 2 1 1","2 0 1 0","1 0 -1 -1"]
assert compile(st,"\n".join(junk).encode())==0; n+=1
class Cfg(C.Structure): _fields_=[("a",C.c_float),("b",C.c_int*3),("c",C.c_float),("d",C.c_int*4)]
L.dcsg_parse_export_config.argtypes=[C.c_char_p,C.c_void_p]
for t in (b"",b"1\n2",b"10.0\n6\n8\n9\n0.7\n50\n16\n512\n4096\n",b"x\ny\n",b"1e400\n-5\n999999999999\n1\n1\n1\n1\n1\n1"):
    cfg=Cfg(); L.dcsg_parse_export_config(t,C.byref(cfg)); n+=1
tab=open("/root/repo/designcsg_amd/data/lookupTable.txt","rb").read()
L.dcsg_expand_lookup_table.argtypes=[C.c_char_p,C.c_void_p,C.POINTER(C.c_int)]
out=(C.c_ubyte*4096)(); tot=C.c_int()
assert L.dcsg_expand_lookup_table(tab,out,C.byref(tot))==0
for cut in (0,10,100,1000,len(tab)-1):
    L.dcsg_expand_lookup_table(tab[:cut]+b"\n99 99 99 zz",out,C.byref(tot)); n+=1
rng=np.random.default_rng(0)
for f,extra in (("dcsg_write_stl",False),("dcsg_write_ply",False),("dcsg_write_stl_normals",False),("dcsg_write_ply_indexed",True)):
    fn=getattr(L,f); fn.argtypes=[C.c_char_p,C.c_void_p,C.c_uint64]+([C.POINTER(C.c_uint64)] if extra else [])
    for m in (0,1,33,40000,5):
        t=rng.normal(size=(m,3,3)).astype(np.float32)
        args=[os.path.join(os.path.dirname(os.environ.get("DCSG_ASAN_LIB", "/tmp/dcsg_asan/x")), "o.bin").encode(),t.ctypes.data_as(C.c_void_p),m]+([C.byref(C.c_uint64())] if extra else [])
        assert fn(*args)==0; n+=1
# byte-range writers: three ranges in shuffled order between BEGIN and END, over a longer stale file
for f in ("dcsg_write_stl_range","dcsg_write_ply_range"):
    fn=getattr(L,f); fn.argtypes=[C.c_char_p,C.c_void_p,C.c_uint64,C.c_uint64,C.c_uint64,C.c_uint]
    path=os.path.join(os.path.dirname(os.environ.get("DCSG_ASAN_LIB", "/tmp/dcsg_asan/x")), "r.bin").encode()
    open(path,"wb").write(b"z"*9_000_000)
    t=rng.normal(size=(70000,3,3)).astype(np.float32); cuts=[0,1,30000,70000]
    assert fn(path,None,0,0,70000,1)==0
    for k in (2,0,1):
        part=np.ascontiguousarray(t[cuts[k]:cuts[k+1]])
        assert fn(path,part.ctypes.data_as(C.c_void_p),len(part),cuts[k],70000,0)==0
    assert fn(path,None,0,0,70000,2)==0; n+=5
    assert fn(path,t.ctypes.data_as(C.c_void_p),10,69995,70000,0)!=0
L.dcsg_angle_q_star.restype=C.c_float; L.dcsg_angle_q_star.argtypes=[C.c_float]
for v in (0.0,0.785,3.2,-1.0,float("nan")): L.dcsg_angle_q_star(v)
print("sanitizer run finished:", n, "calls")
