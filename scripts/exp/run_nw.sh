for m in 49152 47104 40960 36864 34816 57344 61440 24576 20480; do
python scripts/exp/trsm32_ab.py scripts/exp/librlhip_old.so $m 2>&1 | grep TFLOP
python scripts/exp/trsm32_ab.py randlapack_amd/librlhip.so $m 2>&1 | grep TFLOP
done
