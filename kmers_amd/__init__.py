"""kmers_amd -- MI355X-native (HIP, gfx950) k-mer shredding and counting behind kmerexpr's own interface.

    from kmers_amd import transcriptome_reader as tr
    x = tr.transcriptome_to_x(K, ISO_FILE, X_FILE)            # reference kmerexpr/transcriptome_reader.py:19
    from kmers_amd.libkmers import fasta_count_kmers
    y = fasta_count_kmers(READS_FILE, K)                      # reference kmerexpr/libkmers.py:86
"""
__all__ = ["libkmers", "transcriptome_reader"]
