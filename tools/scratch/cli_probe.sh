#!/bin/bash
# probe of the default CLI run on ca-AstroPh K=20: where do the milliseconds of the sweep loop go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
zcat $R/tests/golden/graphs/ca-AstroPh.csv.gz > /tmp/astro.csv
run() { # name, env..., -- args
  local name=$1; shift
  d=$(mktemp -d); cd $d
  env "$@" SVINET_TIMING_FILE=$d/t.json $R/svinet_amd/bin/svinet -file /tmp/astro.csv -n 17903 -k 20 -link-sampling $EXTRA > $d/out.txt 2> $d/err.txt
  echo "== $name: $(python -c "import json; t=json.load(open('$d/t.json')); print('sweeps %d in %.3f ms = %.4f ms/sweep, chunks %d reports %d report_host %.3f ms ctor %.3f s final %.3f s' % (t['sweeps'], t['sweeps_s']*1e3, t['sweeps_s']/t['sweeps']*1e3, t['chunks'], t['reports'], t.get('report_host_s',0)*1e3, t['ctor_s'], t['final_files_s']))")"
  grep "^\[loop\]" $d/err.txt | head -40
  cd /; rm -rf $d
}
for i in 1 2; do run default_$i A=1; done
run trace SVINET_TRACE_LOOP=1
run graph_after_0 SVILS_GRAPH_AFTER=0
run graph_after_0_trace SVILS_GRAPH_AFTER=0 SVINET_TRACE_LOOP=1
EXTRA="-no-stop -max-iterations 299" run long300 A=1
