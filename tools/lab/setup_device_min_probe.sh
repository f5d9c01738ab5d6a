for dm in 20000 5000 1500; do
  echo "== FEMSHELL_AMG_DEVICE_MIN=$dm"
  FEMSHELL_AMG_DEVICE_MIN=$dm FEMSHELL_AMG_VERBOSE=1 python3 tools/amg_probe.py panel 1414 2>&1 | grep -E "level [123] |wall_s" | cut -c1-150 | grep -v "^   {" | head -40
done
