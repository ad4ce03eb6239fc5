#!/usr/bin/env python3
"""hair_retrieval.py — the retrieval CLI of src/hair_retrieval.py on the MI355X hot path.

Same flags (src/hair_retrieval.py:8-57) and the same flow (:82-147,150-190): build the HairEncoder, extract
or load `embeddings.npy` + `image_paths.txt`, encode the query image, print the top-k.  The embed runs through
hcir.vit_engine (HIP), the cosine top-k through hcir_sim_topk against the gallery resident in HBM.
`--save_visualization` (matplotlib collages, HairRetrievalVisualizer) is outside the hot path: the flag is
accepted and the retrieval results it would draw are printed instead.
"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from hcir.hair_encoder import HairEncoder


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Hair Image Retrieval Inference")
    parser.add_argument("--ckpt_path", type=str, default=None, help="Path to model checkpoint")
    parser.add_argument("--model_name", type=str, default="vit_base_patch16",
                        choices=["vit_base_patch16", "sim_vit_base_patch16"], help="Model architecture to use")
    parser.add_argument("--data_path", type=str, default="data/train", help="Path to training data directory")
    parser.add_argument("--batch_size", type=int, default=64, help="Batch size for inference")
    parser.add_argument("--num_workers", type=int, default=8, help="Number of workers for data loading")
    parser.add_argument("--device", type=str, default=None, choices=["cuda", "cpu"],
                        help="Device to use (cuda/cpu). If None, auto-detect")
    parser.add_argument("--embed_save_dir", type=str, default="save/embeddings", help="Directory to save embeddings")
    parser.add_argument("--query_image", type=str, default=None,
                        help="Path to query image for retrieval (if None, use first image from dataset)")
    parser.add_argument("--top_k", type=int, default=5, help="Number of top similar images to retrieve")
    parser.add_argument("--num_queries", type=int, default=5,
                        help="Number of random query images to use for visualization")
    parser.add_argument("--save_visualization", action="store_true", help="Save retrieval visualizations")
    parser.add_argument("--vis_save_dir", type=str, default="save/visualizations",
                        help="Directory to save visualizations")
    parser.add_argument("--random_seed", type=int, default=42, help="Random seed for reproducible query selection")
    parser.add_argument("--extract_only", action="store_true", help="Only extract embeddings, skip retrieval")
    parser.add_argument("--retrieve_only", action="store_true",
                        help="Only perform retrieval, skip embedding extraction")
    parser.add_argument("--force_extract", action="store_true",
                        help="Force re-extraction of embeddings even if they exist")
    return parser.parse_args(argv)


def extract_embeddings(hair_encoder, args):
    print("Extracting embeddings from dataset...")
    return hair_encoder.extract_dataset_features(data_path=args.data_path, batch_size=args.batch_size,
                                                 num_workers=args.num_workers, save_dir=args.embed_save_dir)


def load_embeddings(hair_encoder, args):
    print("Loading existing embeddings...")
    return hair_encoder.load_embeddings(args.embed_save_dir)


def single_query_retrieval(hair_encoder, embeddings, paths, args):
    if args.query_image:
        query_img_path = args.query_image
    else:
        query_img_path = paths[0]
        print(f"No query image specified, using first image from dataset: {query_img_path}")
    print(f"Encoding query image: {query_img_path}")
    query_embedding = hair_encoder.encode_single_image(query_img_path)
    results = hair_encoder.retrieve_similar_images(query_embedding, embeddings, paths, top_k=args.top_k)
    print(f"\nTop {args.top_k} similar images to: {os.path.basename(query_img_path)}")
    print("-" * 60)
    for i, res in enumerate(results):
        print(f"{i+1}. {os.path.basename(res['path'])} (similarity: {res['similarity']:.4f})")
    return query_img_path, results


def multiple_query_retrieval(hair_encoder, embeddings, paths, args):
    """The queries HairRetrievalVisualizer.visualize_multiple_queries would draw (random.seed + random.sample
    over the stored paths), results printed instead of plotted."""
    random.seed(args.random_seed)
    out = []
    for qp in random.sample(paths, min(args.num_queries, len(paths))):
        qe = hair_encoder.encode_single_image(qp)
        res = hair_encoder.retrieve_similar_images(qe, embeddings, paths, top_k=args.top_k + 1)
        res = [r for r in res if r["path"] != qp][: args.top_k]     # the query itself is skipped in the collage
        print(f"{os.path.basename(qp)} -> " + ", ".join(f"{os.path.basename(r['path'])} ({r['similarity']:.4f})" for r in res))
        out.append((qp, res))
    return out


def main(argv=None):
    args = parse_args(argv)
    print("Initializing HairEncoder...")
    hair_encoder = HairEncoder(ckpt_path=args.ckpt_path, model_name=args.model_name, device=args.device)
    should_extract = not args.retrieve_only and (args.force_extract
                                                 or not hair_encoder.check_embeddings_exist(args.embed_save_dir))
    embeddings = paths = None
    if should_extract:
        print("Force extraction enabled." if args.force_extract else "Embeddings not found.")
        embeddings, paths = extract_embeddings(hair_encoder, args)
    elif not args.extract_only:
        embeddings, paths = load_embeddings(hair_encoder, args)
    result = None
    if not args.extract_only:
        if args.save_visualization:
            print("(visualisation collages are outside the hot path: printing the retrievals)")
            result = multiple_query_retrieval(hair_encoder, embeddings, paths, args)
        else:
            result = single_query_retrieval(hair_encoder, embeddings, paths, args)
    print("\n" + "=" * 60 + "\nINFERENCE COMPLETED\n" + "=" * 60)
    return result


if __name__ == "__main__":
    main()
