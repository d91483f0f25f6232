# the evaluation cache's verify mode over whole games at C3's size: random-init and peaked priors, lock-step and refill
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_verify_eval_cache_whole_games.txt
timeout -k 10 300 python tools/verify_eval_cache.py 70 16384 > $O 2>&1
timeout -k 10 300 python tools/verify_eval_cache.py 70 16384 peaked >> $O 2>&1
timeout -k 10 400 python tools/verify_eval_cache.py 70 16384 peaked refill >> $O 2>&1
grep -v amdgpu.ids $O
