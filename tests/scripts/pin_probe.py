"""Measurement probe (not a test): cost of allocating / first-touching / freeing a 1.3 GB host block in the ways the host path of
ommCpuBake could obtain its result memory, and the D2H rate into each.  Result on the MI355X box (DESIGN.md fences, HostPool in
omm_host.cpp): fresh malloc = 72-103 ms first D2H (page faults) + 95 ms free; warm pages = 24 ms (57 GB/s); hipHostMalloc = 235 ms."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
n = 1300 * 1024 * 1024
d = C.c_void_p(); assert hip.hipMalloc(C.byref(d), C.c_size_t(n)) == 0
hip.hipMemset(d, 1, C.c_size_t(n)); hip.hipDeviceSynchronize()
for flags in (0, 0x2):  # default, hipHostMallocMapped? (0x2)
    t=time.time(); p = C.c_void_p(); r = hip.hipHostMalloc(C.byref(p), C.c_size_t(n), C.c_uint(flags)); t1=time.time()
    hip.hipMemcpy(p, d, C.c_size_t(n), 2); t2=time.time()
    hip.hipMemcpy(p, d, C.c_size_t(n), 2); t3=time.time()
    hip.hipHostFree(p); t4=time.time()
    print("flags",flags,"rc",r,"alloc %.1f ms, d2h first %.1f ms (%.1f GB/s), second %.1f ms, free %.1f ms"%((t1-t)*1e3,(t2-t1)*1e3,n/(t2-t1)/1e9,(t3-t2)*1e3,(t4-t3)*1e3))
libc = C.CDLL("libc.so.6"); libc.malloc.restype = C.c_void_p; libc.malloc.argtypes=[C.c_size_t]; libc.free.argtypes=[C.c_void_p]
for i in range(2):
    t=time.time(); p = libc.malloc(n); t1=time.time(); hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(n), 2); t2=time.time(); hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(n), 2); t3=time.time(); libc.free(p); t4=time.time()
    print("malloc %.1f ms, d2h first %.1f ms (%.1f GB/s), second %.1f ms, free %.1f ms"%((t1-t)*1e3,(t2-t1)*1e3,n/(t2-t1)/1e9,(t3-t2)*1e3,(t4-t3)*1e3))
# hipHostRegister of malloc'd memory
p = libc.malloc(n); C.memset(C.c_void_p(p), 0, n)
t=time.time(); r=hip.hipHostRegister(C.c_void_p(p), C.c_size_t(n), 0); t1=time.time(); hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(n), 2); t2=time.time(); hip.hipHostUnregister(C.c_void_p(p)); t3=time.time()
print("register rc",r,"%.1f ms, d2h %.1f ms, unregister %.1f ms"%((t1-t)*1e3,(t2-t1)*1e3,(t3-t2)*1e3))
