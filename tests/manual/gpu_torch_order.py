import sys, os, time
t=time.time(); import torch; print("torch import s", time.time()-t, torch.__version__)
sys.argv=[sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)),"gpu_first.py")).read())
import subprocess
print(subprocess.run("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u"%os.getpid(),shell=True,capture_output=True,text=True).stdout)
torch.cuda.synchronize(); print("torch sync ok", torch.cuda.get_device_name(0))
