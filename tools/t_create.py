#!/usr/bin/env python3
"""tools/t_create.py — dlopen of the engine library, first and second yacrd_engine_create (HIP start-up): GPU box."""
import time, sys, os
t0=time.perf_counter()
sys.path.insert(0,'/root/repo')
import ctypes
lib = ctypes.CDLL('/root/repo/yacrd_amd/lib/libyacrd_hip.so')
t1=time.perf_counter()
class Cfg(ctypes.Structure): _fields_=[("device_id",ctypes.c_int32),("flags",ctypes.c_uint32)]
e=ctypes.c_void_p(); cfg=Cfg(0,0)
rc=lib.yacrd_engine_create(ctypes.byref(cfg), ctypes.byref(e))
t2=time.perf_counter()
e2=ctypes.c_void_p()
rc=lib.yacrd_engine_create(ctypes.byref(cfg), ctypes.byref(e2))
t3=time.perf_counter()
print("dlopen %.1f ms, first engine_create %.1f ms (rc %d), second %.1f ms" % ((t1-t0)*1e3,(t2-t1)*1e3,rc,(t3-t2)*1e3))
