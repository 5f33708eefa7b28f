"""Tevatron ranking jsonl -> neighbor JSON (reference: retrieve/convert_format.py:3-16, which
hard-codes its two paths; here they are arguments)."""
import argparse

from .neighbors import convert_tevatron_file


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('input_path', help='Tevatron *.jsonl with query_id / negative_passages[].docid')
    ap.add_argument('output_path', help='neighbor file, e.g. contrastive/test.json')
    a = ap.parse_args(argv)
    n = convert_tevatron_file(a.input_path, a.output_path)
    print(f"{n} queries -> {a.output_path}")
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
