for v in "" linear groupnorm matvec "linear,groupnorm,matvec"; do
  echo "== TORCH=$v"
  HECKA_SAMPLER_TORCH=$v HECKA_YARD_REPORT=1 timeout -k 10 200 python -u -m pytest tests/test_hypernet.py -m gpu -q --timeout 240 --timeout-method thread -k "train_step_matches" -s 2>&1 | grep YARD | sed 's/  */ /g' | sort -k5 -g | awk '{a[$2]=$5" "$3} END{for(k in a) print k, a[k]}'
done
