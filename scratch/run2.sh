cd /root/repo
for e in 0 1 2 3 4 5 7 8 15; do EXPLIB=/root/repo/scratch/exp/lib$e.so python scratch/exp.py 2>&1 | grep -v amdgpu | tail -1; done
