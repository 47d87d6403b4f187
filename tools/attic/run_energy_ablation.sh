mkdir -p gpurun_out/r3l
ARMS="presplit" timeout 120 python tools/attic/dbg_power.py > gpurun_out/r3l/abl.txt 2>&1
for v in 1 2 3 4 8 11 15 7; do LOCOV_HIP_LIB=tools/liblocov_abl$v.so ARMS="presplit" timeout 120 python tools/attic/dbg_power.py >> gpurun_out/r3l/abl.txt 2>&1; done
grep "|" gpurun_out/r3l/abl.txt
