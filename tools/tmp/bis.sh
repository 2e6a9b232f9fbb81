cd $GRAFT_REPO_ROOT
for d in tools/tmp/wt_*; do
  echo "== $d"
  cp tools/tmp/big.py $d/big.py
  sed -i "s#/root/repo#$GRAFT_REPO_ROOT/$d#" $d/big.py
  (cd $d && timeout -k 10 120 python big.py 2>&1 | grep -v amdgpu | head -3)
done
