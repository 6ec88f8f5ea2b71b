cd $GRAFT_REPO_ROOT
export CDKF_RTC_CACHE_DIR=$GRAFT_REPO_ROOT/gpurun_out/rtc_cache_tmp; mkdir -p $CDKF_RTC_CACHE_DIR; chmod 755 $CDKF_RTC_CACHE_DIR
for pol in o3 o3basic; do for uni in 0 1; do
  echo "=== policy $pol uniform $uni"
  CDKF_RTC_POLICY=$pol CDKF_RTC_UNIFORM=$uni timeout 900 python scripts/r5_o3_probe.py case d2grad 2>&1 | tail -1 | cut -c1-200
  CDKF_RTC_POLICY=$pol CDKF_RTC_UNIFORM=$uni timeout 900 python -m pytest tests/test_custom_drift.py -m gpu -q -x -k "derivatives_by_dual_numbers or forward_sens" 2>&1 | tail -3 | cut -c1-300
done; done
