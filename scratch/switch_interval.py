"""scratch/switch_interval.py SECONDS [bench args]: bench.py with the interpreter's thread switch interval set first."""
import runpy
import sys

sys.setswitchinterval(float(sys.argv[1]))
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path("bench.py", run_name="__main__")
