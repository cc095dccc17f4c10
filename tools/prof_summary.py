import csv, sys
path = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 8
rows=list(csv.DictReader(open(path)))
cat={}; calls={}
def c(name):
    if name.startswith('Cijk'): return 'gemm(hipblaslt)'
    for k, lab in (('wgrad256', 'grouped wgrad GEMM (wgrad256 / wgrad_lw)'), ('wgrad_lw', 'grouped wgrad GEMM (wgrad256 / wgrad_lw)'), (' ln_', 'layernorm (ln_/pm_ln_)'), ('pm_ln', 'layernorm (ln_/pm_ln_)'), (' gn_', 'groupnorm (gn_)'), ('gelu_', 'gelu'), ('colsum', 'colsum (bias grads)'), ('residual_', 'residual add'), ('detic_', 'fused head losses'), ('cn_loss', 'fused head losses'), ('cascade_refine', 'cascade_refine'), ('wgrad_gemm', 'wgrad 128 kernel')):
        if k in (' ' + name): return lab
    for k in ('win_attn_bwd','win_attn_fwd','nms_sweep','nms_mask','roi_align_sep_kernel<unsigned short, true','roi_align_sep_kernel<unsigned short, false','roi_align','adamw','im2col','col2im','window_shuffle','cp_','mask_crop','centernet_targets','iou_match','gemm_'):
        if k in name: return k
    if 'layer_norm' in name or 'GammaBeta' in name or 'cuComputeGradInput' in name or 'RowwiseMoments' in name: return 'layernorm/groupnorm'
    if 'copy_kernel' in name or 'copyBuffer' in name: return 'casts/copies'
    if 'fillBuffer' in name or 'FillFunctor' in name: return 'fills'
    if 'elementwise' in name: return 'elementwise'
    if 'reduce_kernel' in name: return 'reduce'
    return 'other'
for r in rows:
    k=c(r['Name']); cat[k]=cat.get(k,0)+int(r['TotalDurationNs']); calls[k]=calls.get(k,0)+int(r['Calls'])
tot=sum(cat.values())
print("GPU busy %.1f ms/step, %d launches/step" % (tot/1e6/steps, sum(calls.values())/steps))
for k,v in sorted(cat.items(), key=lambda x:-x[1]):
    print("%-45s %8.2f ms/step  %6d calls/step" % (k, v/1e6/steps, calls[k]/steps))
