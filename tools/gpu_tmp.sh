cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
python tools/shard_cost.py astroph-k200 2,4,8 2>/dev/null | tee $O/shard_cost_config4.txt
python tools/shard_cost.py mmsb:1000000:512:24 2,4,8 2>/dev/null | tee $O/shard_cost_config5.txt
