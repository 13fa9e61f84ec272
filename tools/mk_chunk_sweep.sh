for c in 1024 2048 3072 4096 6144 8192 16384; do
  echo "== OBB_NMS_MK_CHUNK=$c"
  env OBB_NMS_MK=1 OBB_NMS_MK_CHUNK=$c python tools/mk_time.py $REGIMES 2>&1 | grep -v amdgpu
done
