"""Runs bench.py in this process and writes /proc/self/maps at interpreter exit, so that the addresses of a crash inside
exit() can be attributed to libraries.  usage: python3 tools/exit_probe.py MAPS_FILE [nofix] [bench.py arguments...]
nofix: unregister the package's atexit shutdown (handles are then freed by __del__ during finalisation, as before)."""
import atexit
import os
import runpy
import sys

sys.path.insert(0, os.getcwd())
maps_file = sys.argv[1]
args = sys.argv[2:]
nofix = bool(args) and args[0] == "nofix"
if nofix:
    args = args[1:]


def dump():
    with open(maps_file, "w") as fp, open("/proc/self/maps") as src:
        fp.write(src.read())


atexit.register(dump)
if nofix:
    from compression_amd import _lib
    _lib.load()
    atexit.unregister(_lib.shutdown)
sys.argv = ["bench.py"] + args
runpy.run_path("bench.py", run_name="__main__")
