"""Turn the JSON lines the GPU parity tests append to gpurun_out/parity.jsonl (tests/test_gpu_configs.py::_record: per-tensor
gradient errors against the float64 oracle on the HIP run's branch with the free float64 run's and the fp32 CPU oracle's
distances beside them, the bf16 figures against the bf16-operand oracle, the teacher-forced per-launch errors) into ONE
committed file, profiles/parity_rNN.json, so that the gates can be audited without a GPU:
    python tools/parity_report.py gpurun_out/parity.jsonl profiles/parity_r04.json"""
import json
import sys


def main(src, dst):
    records = {}
    for line in open(src):
        line = line.strip()
        if line:
            r = json.loads(line)
            records[r.pop('test')] = r                      # the last run of a test wins
    summary = {}
    for name, r in records.items():
        if 'tensors' in r:
            w = r['worst'][0]
            summary[name] = (f"{r['within_tol_outright']} of {r['tensors']} tensors within {r['tol']:g} outright, {r['failed']} failed; "
                             f"worst {w['name']} {w['err']:.2e} on the HIP run's branch (free float64 run "
                             f"{w.get('err_vs_free_float64_run', float('nan')):.2e}, fp32 CPU oracle vs free float64 "
                             f"{w.get('fp32_cpu_oracle_vs_free_float64', float('nan')):.2e}); "
                             f"{r.get('positions_the_free_float64_run_decides_differently', '?')} positions decided differently")
        elif 'quantities' in r:
            summary[name] = (f"{r['quantities']} quantities ({r.get('launches_with_bf16_operands', '-')} launches with bf16 operands) within "
                             f"{r['tol']:g}; worst {r['worst'][0]['name']} {r['worst'][0]['err']:.2e}")
        else:
            summary[name] = ', '.join(f'{k} {v:.2e}' for k, v in r.items() if isinstance(v, float))
    json.dump({'summary': summary, 'records': records}, open(dst, 'w'), indent=1, sort_keys=True)
    for k, v in sorted(summary.items()):
        print(f'{k}: {v}')


if __name__ == '__main__':
    main(*sys.argv[1:3])
