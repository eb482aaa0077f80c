# kernel-tuning variants of the same HIP sources (built with -D switches into fv3_jedi_linearmodel_amd/variants/): one short bench each
for v in default tl512 tl512nl256 nowrite; do
  unset FV3LM_LIB FV3LM_NO_AD_WRITE_MODE
  if [ $v = nowrite ]; then export FV3LM_NO_AD_WRITE_MODE=1; elif [ $v != default ]; then export FV3LM_LIB=$PWD/fv3_jedi_linearmodel_amd/variants/$v.so; fi
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --profile-out gpurun_out/r2_var_$v.txt > gpurun_out/r2_var_$v.json 2> gpurun_out/r2_var_$v.err
  echo "== $v: $(python -c "import json;print(json.load(open('gpurun_out/r2_var_$v.json'))['ms_per_step'])")"; grep -E "TpFused|TpPpmY.ad|TpPpmX.ad|CswTransport.ad|fill|memset" gpurun_out/r2_var_$v.txt
done
