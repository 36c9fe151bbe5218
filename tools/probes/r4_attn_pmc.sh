cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4attnpmc; mkdir -p $O
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in ${VARS:-0}; do
export VITAE_ATTN_FWD=$v
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/p1 -- python tools/attn_one.py 4 1729 16 32 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/p2 -- python tools/attn_one.py 4 1729 16 32 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_FLAT --output-format csv -d $O/p3 -- python tools/attn_one.py 4 1729 16 32 > $O/p3.log 2>&1
python tools/summarize_pmc.py $O/p1 $O/p2 $O/p3 > $O/attn_pmc_v$v.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3
echo "== var $v"; grep -i "attn_fwd" $O/attn_pmc_v$v.txt | cut -c1-700
done
