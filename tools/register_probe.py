"""Diagnostic: which host buffers hipHostRegister (through qcnn_host_register) accepts on this machine."""
import ctypes as C, mmap, os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcnn_gpu_amd import capi
lib = capi.load()
n = C.c_int(); capi.check(lib.qcnn_device_count(C.byref(n))); capi.check(lib.qcnn_set_device(0))
print("memlock limit", resource.getrlimit(resource.RLIMIT_MEMLOCK))
def tryreg(name, addr, size):
    rc = lib.qcnn_host_register(C.c_void_p(addr), size)
    print(f"{name}: addr%4096={addr % 4096} size={size} rc={rc} {lib.qcnn_last_error().decode() if rc else ''}")
    if rc == 0: print("   unregister rc", lib.qcnn_host_unregister(C.c_void_p(addr)))
a = np.empty(107520, np.uint8); tryreg("numpy small", a.ctypes.data, a.nbytes)
b = np.empty(8 << 20, np.uint8); tryreg("numpy 8 MiB", b.ctypes.data, b.nbytes)
m = mmap.mmap(-1, 1 << 20); mb = (C.c_char * (1 << 20)).from_buffer(m); tryreg("mmap 1 MiB", C.addressof(mb), 1 << 20)
m2 = mmap.mmap(-1, 64 << 20); mb2 = (C.c_char * (64 << 20)).from_buffer(m2); m2[:] = b"\1" * (64 << 20); tryreg("mmap 64 MiB touched", C.addressof(mb2), 64 << 20)
d = C.c_void_p(); capi.check(lib.qcnn_dev_alloc(C.byref(d), 1 << 20))
tryreg("numpy small after hipMalloc", a.ctypes.data, a.nbytes)
