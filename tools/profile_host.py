"""Host-side profile of Trainer.step on the GPU box: per-step enqueue times and a cProfile top list."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    device = torch.device('cuda:0')
    torch.cuda.set_device(device)
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.trainer import Trainer
    torch.manual_seed(0)
    model = weak_label.CRNN.build().to(device)
    trainer = Trainer(model, lr=5e-4, gradient_clipping=1e10)
    batch = bench.synth_batch(32, device)
    for _ in range(3):
        trainer.step(batch)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); trainer.step(batch); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); torch.cuda.synchronize(); tail = (time.perf_counter() - t0) * 1e3
    print('enqueue ms per step:', [round(t, 2) for t in ts], 'final sync wait ms:', round(tail, 2))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        trainer.step(batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(28)

if __name__ == '__main__':
    main()
