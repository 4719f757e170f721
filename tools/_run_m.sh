bash tools/pmc_mfma.sh r04
echo done
