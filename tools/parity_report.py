"""Turn the JSON lines the GPU parity tests append to gpurun_out/parity.jsonl (tests/test_gpu_configs.py::_record: per-tensor
gradient errors against the float64 oracle with the rounding sensitivity and the fp32 CPU oracle's own error beside them, the
measured bf16 logit / score / loss / gradient errors) into ONE committed file, profiles/parity_rNN.json, so that the gates'
escape hatches can be audited without a GPU:   python tools/parity_report.py gpurun_out/parity.jsonl profiles/parity_r03.json"""
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
            summary[name] = (f"{r['within_tol_outright']} of {r['tensors']} tensors within {r['tol']:g} outright, {r['failed']} failed; "
                             f"worst {r['worst'][0]['name']} {r['worst'][0]['err']:.2e} (rounding sensitivity "
                             f"{r['worst'][0]['rounding_sensitivity']:.2e}, fp32 CPU oracle {r['worst'][0]['fp32_cpu_oracle_err']:.2e})")
        else:
            summary[name] = ', '.join(f'{k} {v:.2e}' for k, v in r.items() if isinstance(v, float))
    json.dump({'summary': summary, 'records': records}, open(dst, 'w'), indent=1, sort_keys=True)
    for k, v in sorted(summary.items()):
        print(f'{k}: {v}')


if __name__ == '__main__':
    main(*sys.argv[1:3])
