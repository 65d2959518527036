set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
./deep-fusion_amd/tools/probe/probe_mfma_war 200000 > gpurun_out/r2/probe_mfma_war.json 2>&1
cat gpurun_out/r2/probe_mfma_war.json
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2/pytest_gpu.log 2>&1; echo "pytest rc $?"
tail -5 gpurun_out/r2/pytest_gpu.log
timeout -k 10 120 python profiles/stamps.py u8 > gpurun_out/r2/stamps_u8.txt 2>&1
timeout -k 10 120 python profiles/stamps.py s32 > gpurun_out/r2/stamps_s32.txt 2>&1
timeout -k 10 600 bash profiles/collect_pmc.sh r2base_res2a_u8 --workload res2a --dst u8 > gpurun_out/r2/pmc_u8.log 2>&1
cat gpurun_out/r2/stamps_u8.txt
