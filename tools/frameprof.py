import cProfile, pstats, sys, os, io
sys.argv = ['framebench', '--frames', '1']
ROOT = '/root/repo'
sys.path.insert(0, os.path.join(ROOT, 'tools'))
src = open(os.path.join(ROOT, 'tools', 'framebench.py')).read()
# split: setup part and the frame loop
setup, loop = src.split('for f in range(a.frames + 1):')[0], None
exec(compile(setup, 'fb_setup', 'exec'))
import torch, time
def frame():
  smp = sample_ray.RaySamplerSingleImage(data, dev)
  rb = smp.get_all()
  ret = render_image.render_single_image_nvi((fidx, None), (temb, None), (toff, None), smp, rb, model, proj, a.chunk, 64, args, inv_uniform=True,
                                             N_importance=64, det=True, coarse_featmaps=cfeat, fine_featmaps=ffeat, is_train=False)
  torch.cuda.synchronize()
  return ret
frame(); frame()
t0 = time.perf_counter(); frame(); print('frame ms', 1e3 * (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); frame(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:6000])
