"""Seconds to load a database from its files (run on the GPU box): mc_load_db on one context, or mc_group_load_db with
several members (MC_GROUP_DEVICES=0,0,0 rehearses N members on one card; --mode shards cuts the table into N parts).
Plain ctypes on purpose -- only the entry points every build of the library has -- so that --lib can point at an older
build (before / after of the loader's pipeline).
    python tools/load_time.py <base> [--lib libmcclark.so] [--members N] [--mode auto|replicas|shards] [--k 31] [--htsize H]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("base")
ap.add_argument("--lib", default=os.path.join(ROOT, "jn_cuclark_amd", "libmcclark.so"))
ap.add_argument("--members", type=int, default=1)
ap.add_argument("--mode", default="auto")
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--htsize", type=int, default=1610612741)
ap.add_argument("--targets", type=int, default=4096)
ap.add_argument("--reps", type=int, default=1)
a = ap.parse_args()
if a.members > 1:
    os.environ["MC_GROUP_DEVICES"] = ",".join(["0"] * a.members)
lib = C.CDLL(a.lib)
lib.mc_last_error.restype = C.c_char_p
vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int


def chk(rc, what):
    if rc != 0:
        raise SystemExit("%s: %d %s" % (what, rc, lib.mc_last_error().decode()))


for rep in range(a.reps):
    t0 = time.time()
    if a.members == 1:
        lib.mc_open.argtypes = [C.POINTER(vp), i32, u32, u64, u32, u32]
        lib.mc_load_db.argtypes = [vp, C.c_char_p, i32, u32, u64, u64]
        lib.mc_close.argtypes = [vp]
        h = vp()
        chk(lib.mc_open(C.byref(h), 0, a.k, a.htsize, a.targets, 15), "mc_open")
        t1 = time.time()
        chk(lib.mc_load_db(h, a.base.encode(), 4, 1, 0, 0), "mc_load_db")
        dt = time.time() - t1
        lib.mc_close(h)
    else:
        lib.mc_group_open.argtypes = [C.POINTER(vp), vp, u32, u32, u64, u32, u32]
        lib.mc_group_load_db.argtypes = [vp, C.c_char_p, i32, u32, i32]
        lib.mc_group_close.argtypes = [vp]
        g = vp()
        chk(lib.mc_group_open(C.byref(g), None, 0, a.k, a.htsize, a.targets, 15), "mc_group_open")
        t1 = time.time()
        chk(lib.mc_group_load_db(g, a.base.encode(), 4, 1, {"auto": 0, "replicas": 1, "shards": 2}[a.mode]), "mc_group_load_db")
        dt = time.time() - t1
        lib.mc_group_close(g)
    print("%s: %d member(s), mode %s, rep %d: loaded in %.2f s (open %.2f s)" % (os.path.basename(a.lib), a.members, a.mode, rep, dt, t1 - t0), flush=True)
