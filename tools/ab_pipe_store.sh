# DEVELOPER-ONLY: result-store policies of k_mix_dec_mfma (PEBBLEGPU_BANK_DBG 128 nontemporal / 256 sc0 / 512 sc0 sc1) under two-stage calls
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for dbg in 0 128 256 512; do
  PEBBLEGPU_BANK_DBG=$dbg python tools/ab_bank_pipe.py 2 400 2>&1 | tail -1 | sed "s/^/dbg=$dbg /"
done
done
