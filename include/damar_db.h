/* damar_db.h -- in-memory read-block layout consumed by the overlap hot path.
 *
 * Binary-compatible with the reference's HITS_DB / HITS_READ / HITS_TRACK
 * (reference db/DB.h:319-389) so that a block loaded by the reference's own
 * Open_DB + Read_All_Sequences (db/DB.c:457-680, 1547-1608) can be handed to
 * this library unchanged, and a block loaded by damar_read_block() can be handed
 * to the reference.  Only the subset of the DB API that dalign/daligner.c touches
 * is provided (SURVEY.md section 2, row db/DB.c).
 */
#ifndef DAMAR_DB_H
#define DAMAR_DB_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t  int64;
typedef uint64_t uint64;
typedef uint32_t uint32;
typedef uint16_t uint16;
typedef uint8_t  uint8;
typedef int16_t  int16;

#define DB_QV    0x03ff
#define DB_CSS   0x0400
#define DB_BEST  0x0800

/* reference db/DB.h:319-326 (32 bytes: rlen@0 boff@8 coff@16 flags@24) */
typedef struct
{ int    rlen;
  int64  boff;
  int64  coff;
  int    flags;
} HITS_READ;

/* reference db/DB.h:336-342 */
typedef struct _track
{ struct _track *next;
  char          *name;
  int            size;
  void          *anno;
  void          *data;
} HITS_TRACK;

/* reference db/DB.h:361-389 (88 bytes) */
typedef struct
{ int         ureads;
  float       freq[4];
  int         maxlen;
  int64       totlen;
  int         nreads;
  int         part;
  int         ufirst;
  char       *path;
  int         loaded;
  void       *bases;
  HITS_READ  *reads;
  HITS_TRACK *tracks;
} HITS_DB;

/* Load block "<root>.<n>" (or a whole unsplit DB "<root>") the way
 * daligner.c:442-509 read_DB does without mask tracks: stub + .idx + all bases
 * unpacked to one byte per base (0..3), read i at reads[i].boff, a 4 before the
 * first read and after every read.  Returns 0, or -1 with a message on stderr. */
int   damar_read_block(const char *name, HITS_DB *block);
void  damar_close_block(HITS_DB *block);

/* A block as it lies in the .bps file: the stretch of the file that holds its reads (2 bits per base, four bases per byte,
 * first base in the top bits, every read padded to a byte: db/DB.h:292 COMPRESSED_LEN, db/DB.c:334-354 Compress_Read) and
 * where each read starts in it.  damar_read_block_packed fills one INSTEAD of unpacking the bases (block->bases stays
 * NULL, block->reads[i].boff are the offsets the unpacked block would have): a quarter of the bytes to read, to keep
 * and to send to the GPU, which unpacks -- and reverse-complements -- them itself (damar_block_upload_packed,
 * include/damar_hip.h).  Returns 0 (packed), 1 (the reads of this block do not lie back to back in the file: the block
 * was read with damar_read_block's unpacking path instead, pk->raw == NULL) or -1. */
typedef struct
{ unsigned char *raw;      /* the stretch */
  int64          nraw;
  uint32        *foff;     /* [nreads] byte offset of read i in the stretch */
  int64          serial;   /* a number no other packed block of this process has (what a staged copy is known by) */
} damar_packed;

int   damar_read_block_packed(const char *name, HITS_DB *block, damar_packed *pk);
void  damar_free_packed(damar_packed *pk);
/* read r of the block, one byte per base, into dst[0 .. rlen) (dst[-1] and dst[rlen] are set to 4): as the unpacked block
 * holds it, or reversed and complemented as the block's complement holds it (daligner.c:511-570) */
void  damar_unpack_read(const damar_packed *pk, const HITS_DB *block, int r, int comp, char *dst);

/* Out-of-place / in-place reverse complement of a loaded block
 * (daligner.c:511-628 complement_DB, mask tracks not supported yet). */
HITS_DB *damar_complement_block(HITS_DB *block, int inplace);

/* The same as a re-entrant copy into *out: own bases and own (mirrored) mask tracks, reads and path shared
 * with `block`; released by damar_free_complement.  The command-line driver prepares blocks ahead with it.
 * A packed block (bases == NULL) gets its frequencies and mask tracks mirrored only. */
void  damar_complement_copy(const HITS_DB *block, HITS_DB *out);
void  damar_free_complement(HITS_DB *c);

/* daligner.c:442-497 (read_DB) + 263-439 (Merge_Size, Merge_Tracks): load the named interval
 * tracks (-m options) of the block and leave their union as the single mask track on
 * block->tracks (anno = int64[nreads+1] in ints, data = [beg,end) pairs), which Sort_Kmers
 * honours (filter.c:474-526) and damar_complement_block mirrors (daligner.c:572-626).
 * Returns 0, -1 on error. */
int damar_load_masks(HITS_DB *block, char **names, int n);

/* "<prefix>" of a path with directory and ".db" suffix removed (db/DB.c Root). */
char *damar_root(const char *name, const char *suffix);
/* Output directory name d%03d_%05d (db/DB.c:1851 getDir). */
char *damar_get_dir(int run, int block);

/* Synthetic input generator = db/simulator.c:100-352 semantics (drand48-exact)
 * piped through FA2db (db/FA2db.c:611-624, 1114-1131) and DBsplit
 * (db/DBsplit.c:201-234): writes <dir>/<root>.db, .<root>.idx, .<root>.bps.
 * Returns the number of blocks, or -1. */
typedef struct
{ double genome_mbp;   /* simulator <genlen>            */
  double coverage;     /* -c (20.)                      */
  double bias;         /* -b (.5)                       */
  int    seed;         /* -r                            */
  int    rmean;        /* -m (10000)                    */
  int    rsdev;        /* -s (2000)                     */
  int    rshort;       /* -x (4000)                     */
  double erate;        /* -e (.15)                      */
  int    block_mbp;    /* DBsplit -s (200)              */
  int    min_len;      /* FA2db -x (1000)               */
  double tandem_frac;  /* fraction of reads that get a tandem array implanted (0: none) */
  int    max_blocks;   /* > 0: stop when this many blocks are complete (the leading blocks of the full DB) */
} damar_sim_params;

void damar_sim_defaults(damar_sim_params *p);
int  damar_sim_write_db(const damar_sim_params *p, const char *dir, const char *root);

#ifdef __cplusplus
}
#endif
#endif
