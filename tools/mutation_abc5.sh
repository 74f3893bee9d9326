#!/bin/bash
# Mutation run for tests/test_aggregator_host.py::test_valid_nested_proofs_with_several_inputs ("a deliberately
# broken ABC_5 in a scratch run turns the new test red").  CPU only.  A scratch copy of the circuit sources gets ONE defect in the
# input accumulator acc = ABC_0 + sum_k x_k ABC_k - the fifth input walks the doubling chain of ABC_4 instead of ABC_5, in every
# form of the accumulator (full generator, registered application's generator, GPU program recorder) - the host translation units
# are rebuilt from it, linked with the tree's device objects into a scratch library, and the valid-nested-proof tests are run
# against that library through ZKHIP_LIB.  Expected: ONLY the k = 9 case of the new test fails (a valid proof's result bit becomes 0);
# the k = 3 case, the one-input tests and every nine-input test that rounds 3-5 had (unrelated key: result bits 0, every
# constraint satisfied; application generator = full generator) still PASS - they could not see this defect.  Nothing in the tree is modified.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${TMPDIR:-/tmp}/zkhip_mutation
rm -rf $W && mkdir -p $W/zecale_amd $W/include $W/obj
cp -r $ROOT/zecale_amd/csrc $W/zecale_amd/csrc
cp $ROOT/include/*.h* $W/include/
H=$W/zecale_amd/csrc/circuit/bls12_377.hpp
# every place that names ABC_(k+1) - the two forms of the accumulator AND the two helpers that pre-compute its slope denominators, so
# that the mutated circuit stays self-consistent (its witness satisfies its own constraints): the defect is SILENT, only the value of
# a valid proof's result bit shows it
before=$(grep -c 'vk.abc\[k + 1\]' $H)
sed -i 's/vk\.abc\[k + 1\]/vk.abc[(k == 4 ? 3 : k) + 1 \/*MUTATION: input 5 uses ABC_4*\/]/g' $H
after=$(grep -c 'MUTATION' $H)
echo "mutated $after of $before lines naming ABC_(k+1) in circuit/bls12_377.hpp"
[ "$after" -ge 4 ]
sed -i 's/abc.push_back({st->vk.abc\[i\].x.value(), st->vk.abc\[i\].y.value()});/abc.push_back({st->vk.abc[i == 5 ? 4 : i].x.value(), st->vk.abc[i == 5 ? 4 : i].y.value()});/' $W/zecale_amd/csrc/aggregator.cpp
grep -c 'i == 5 ? 4 : i' $W/zecale_amd/csrc/aggregator.cpp
for f in aggregator witness_tape; do
  g++ -O2 -std=c++17 -fPIC -pthread -c $W/zecale_amd/csrc/$f.cpp -o $W/obj/$f.o &
done
wait
objs=""
for o in msm ntt qap zkhip_api witness pipeline multi_device; do objs="$objs $ROOT/build/$o.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $W/libzkhip.so $objs $W/obj/aggregator.o $W/obj/witness_tape.o
cd $ROOT
set +e
ZKHIP_LIB=$W/libzkhip.so python -m pytest tests/test_aggregator_host.py -q -p no:cacheprovider \
  -k "valid_nested_proofs_with_several_inputs or nine_inputs_under_an_unrelated_key or witness_and_public_inputs or application_host_generator" 2>&1 | tail -15
echo "pytest exit code with the mutated library: ${PIPESTATUS[0]} (1 = the mutation was caught)"
