#!/bin/bash
# usage: scratch/build_variant.sh NAME "-DSIGGAN_X=1 ..."   -> scratch/libs/NAME.so  (objects in /tmp/siggan_var_NAME)
set -e
NAME=$1; EXTRA=$2
SRC=${SRCDIR:-/root/repo/signature-gan_amd/csrc}
OBJ=/tmp/siggan_var_$NAME; mkdir -p $OBJ /root/repo/scratch/libs
for f in siggan gconv gconv16 ops fc sn mlp; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -ffp-contract=off $EXTRA -c $SRC/$f.hip -o $OBJ/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libs/$NAME.so $OBJ/*.o
ls -la /root/repo/scratch/libs/$NAME.so
