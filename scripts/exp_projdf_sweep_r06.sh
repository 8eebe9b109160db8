for cfg in "SFSN_PDF_FT=16" "SFSN_PDF_FT=32" "SFSN_PDF_FT=16 SFSN_PDF_WGS=512" "SFSN_PDF_FT=32 SFSN_PDF_LDS_KB=76 SFSN_PDF_WGS=512" "SFSN_PDF_FT=16 SFSN_PDF_LDS_KB=76"; do
  echo "== $cfg"; env $cfg python scripts/exp_projdf_r06.py 2>/dev/null | grep "fused=True" | tail -2
done
