#!/bin/bash
# The reference's own toy protocols (experiments/toy_models/speed_and_stability.sh: predict_f over a mesh of training / query
# sizes for Matern32, Matern52, RBF order 6, noise 0.1, SINE signal, float64; experiments/toy_models/mcmc.py: HMC over the
# hyper-parameters) on the HIP backend: python -m pssgp.experiments.toy.  PSSGP = parallel (device), SSGP = sequential (host).
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R/parallel-gps_amd
for cov in Matern32 Matern52 RBF; do
  python -m pssgp.experiments.toy mesh --model PSSGP --cov $cov --mesh-size 4 --n-seeds 5 --noise-variance 0.1
done
python -m pssgp.experiments.toy mesh --model SSGP --cov Matern32 --mesh-size 4 --n-seeds 3 --noise-variance 0.1
for cov in Matern32 Matern52 RBF; do
  for n in 1024 16384; do
    python -m pssgp.experiments.toy hmc --cov $cov --n-training $n --n-samples 300 --n-burnin 100
  done
done
