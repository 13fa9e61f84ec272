for c in 1024 2048 4096 8192 16384; do
  echo "== OBB_NMS_MK_CHUNK=$c"
  env OBB_NMS_MK=1 OBB_NMS_MK_CHUNK=$c python tools/mk_time.py 2>&1 | grep -v amdgpu
done
