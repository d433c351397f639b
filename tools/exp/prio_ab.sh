# A/B of the bench stream configuration on one box: bash tools/exp/prio_ab.sh
run() { echo "$1: $(env $2 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | tail -1 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"; }
for i in 1 2; do
run "chain on the default stream" "S4F_MAIN_PRIORITY=default"
run "chain on a high-priority stream" "S4F_MAIN_PRIORITY=1"
run "chain on a normal-priority stream" "S4F_MAIN_PRIORITY=0"
run "default stream + a 5th stream" "S4F_MAIN_PRIORITY=default S4F_EAGER_STREAM=new"
run "high-priority stream + a 5th stream" "S4F_MAIN_PRIORITY=1 S4F_EAGER_STREAM=new"
done
