for m in 2 0 1 3; do for s in 0 1000 4000; do echo "mode $m stagger $s"; SED_SMODE=$m SED_STAGGER=$s python tools/kbench.py conv --iters 20 2>&1 | grep mfma_fwd; done; done
