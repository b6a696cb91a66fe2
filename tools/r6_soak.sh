# round 6: allocator behaviour of the step over 48 distinct batch shapes (tools/soak_step.py) under allocator settings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_soak; mkdir -p $O
for conf in "default" "expandable_segments:True" "garbage_collection_threshold:0.4" "roundup_power2_divisions:8"; do
  echo "== PYTORCH_HIP_ALLOC_CONF=$conf" | tee -a $O/soak_alloc.txt
  if [ "$conf" = "default" ]; then
    timeout 600 python tools/soak_step.py ${STEPS:-400} 2>&1 | grep -v amdgpu.ids | grep "^step\|soak\|Error\|error" | tee -a $O/soak_alloc.txt
  else
    PYTORCH_HIP_ALLOC_CONF=$conf PYTORCH_CUDA_ALLOC_CONF=$conf timeout 600 python tools/soak_step.py ${STEPS:-400} 2>&1 | grep -v amdgpu.ids | grep "^step\|soak\|Error\|error" | tee -a $O/soak_alloc.txt
  fi
done
