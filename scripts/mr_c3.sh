cd $GRAFT_REPO_ROOT
HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONPATH=$GRAFT_REPO_ROOT MASTER_ADDR=127.0.0.1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29650 tests/multirank_worker.py gpurun_out/mr_c3.npz c3 > gpurun_out/mr_c3.log 2>&1
python - <<'PY'
import numpy as np
r=np.load('gpurun_out/mr_c3.npz')
print('it',r['it'],r['it1'])
print('rel cost diff per iteration:', np.abs(r['cost']-r['cost1'])/np.abs(r['cost1']))
print('ok', r['ok'], r['ok1'])
PY
