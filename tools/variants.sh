for v in default h16t1024 h8t512 h8t256; do
  if [ $v = default ]; then unset FV3LM_LIB; else export FV3LM_LIB=$PWD/fv3_jedi_linearmodel_amd/variants/$v.so; fi
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-out gpurun_out/r2_var_$v.txt > gpurun_out/r2_var_$v.json 2> gpurun_out/r2_var_$v.err
  echo "== $v: $(python -c "import json;print(json.load(open('gpurun_out/r2_var_$v.json'))['ms_per_step'])")"; grep TpFused gpurun_out/r2_var_$v.txt
done
