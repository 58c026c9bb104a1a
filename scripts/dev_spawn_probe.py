"""does a child process (fork+exec) work from a process that has initialised the GPU?"""
import subprocess, torch
torch.cuda.init(); x = torch.ones(4, device="cuda"); print("gpu ok", float(x.sum()))
r = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True, timeout=60)
print("rc", r.returncode, r.stdout.splitlines()[:2], r.stderr[:200])
