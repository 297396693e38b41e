# source me: step NAME SECONDS cmd... -> runs under timeout, logs to $OUT/NAME.log/.err, stops the whole script if the step was killed
step() {
  local name=$1 secs=$2; shift 2
  echo "[$(date +%H:%M:%S)] $name: $*"
  timeout -k 10 $secs "$@" > $OUT/$name.log 2> $OUT/$name.err
  local rc=$?
  echo "[$(date +%H:%M:%S)] $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed at its limit: stopping"; exit 99; fi
  return 0
}
