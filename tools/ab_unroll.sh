# A/B of the pixel loop's shape on one MI355X: unroll factor (pixels evaluated back to back per iteration),
# tile width (columns of a workgroup's 256 work-items) and rows per work-item.
#   usage (from the repo root on the GPU box): bash tools/ab_unroll.sh [workloads...] > gpurun_out/ab_unroll.txt
WL="${@:-ident pond}"
for w in $WL; do
  for tw in 16 32 64; do
    for u in 2 4 8; do
      for ppt in default 16; do
        export MMHIP_TILE_W=$tw MMHIP_UNROLL=$u
        if [ $ppt = default ]; then unset MMHIP_PPT; else export MMHIP_PPT=$ppt; fi
        timeout -k 10 120 python3 bench.py --workload $w --no-extras --steps 20 > /tmp/q.json 2> /tmp/q.err || { echo "$w tile_w=$tw unroll=$u ppt=$ppt FAILED"; tail -3 /tmp/q.err; continue; }
        python3 -c "
import json
j=json.loads(open('/tmp/q.json').read().strip().splitlines()[-1]); print('$w tile_w=$tw unroll=$u ppt=$ppt kernel_ms', round(j['roofline'].get('kernel_ms') or 0,4))"
      done
    done
  done
done
