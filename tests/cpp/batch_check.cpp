// Test program: rmd::SeedMatrixBatch (include/rmd/seed_matrix_batch.cuh) against stand-alone rmd::SeedMatrix objects on the same
// frames, through the C++ drop-in headers with a plain host compiler.  usage: batch_check <in.bin>   (format of facade_check, but with
// n_seq sequences back to back); exit code 0 iff every member's planes equal the stand-alone object's, bit for bit.
#include <cstdio>
#include <cstring>
#include <vector>

#include <rmd/depthmap_denoiser.cuh>
#include <rmd/seed_matrix_batch.cuh>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int hdr[4];
  float K[4], range[2];
  if (fread(hdr, sizeof(int), 4, f) != 4 || fread(K, sizeof(float), 4, f) != 4 || fread(range, sizeof(float), 2, f) != 2) return 2;
  const int n_seq = hdr[0], w = hdr[1], h = hdr[2], n_frames = hdr[3];
  const size_t px = static_cast<size_t>(w) * h;
  std::vector<std::vector<float> > img(static_cast<size_t>(n_seq) * n_frames, std::vector<float>(px));
  std::vector<rmd::SE3<float> > pose(static_cast<size_t>(n_seq) * n_frames);
  for (size_t i = 0; i < img.size(); ++i) {
    float T[12];
    if (fread(&img[i][0], sizeof(float), px, f) != px || fread(T, sizeof(float), 12, f) != 12) return 2;
    for (int k = 0; k < 12; ++k) pose[i].data.data[k] = T[k];
  }
  fclose(f);
  try {
    const rmd::PinholeCamera cam(K[0], K[1], K[2], K[3]);
    rmd::SeedMatrixBatch batch(n_seq, w, h, cam);
    std::vector<rmd::SeedMatrix*> alone;
    for (int s = 0; s < n_seq; ++s) {
      alone.push_back(new rmd::SeedMatrix(w, h, cam));
      batch[s].setReferenceImage(&img[static_cast<size_t>(s) * n_frames][0], pose[static_cast<size_t>(s) * n_frames], range[0], range[1]);
      alone[s]->setReferenceImage(&img[static_cast<size_t>(s) * n_frames][0], pose[static_cast<size_t>(s) * n_frames], range[0], range[1]);
    }
    std::vector<float*> frames(n_seq);
    std::vector<rmd::SE3<float> > T(n_seq);
    for (int k = 1; k < n_frames; ++k) {
      for (int s = 0; s < n_seq; ++s) {
        frames[s] = &img[static_cast<size_t>(s) * n_frames + k][0];
        T[s] = pose[static_cast<size_t>(s) * n_frames + k];
        alone[s]->update(frames[s], T[s]);
      }
      batch.update(&frames[0], &T[0]);
    }
    int bad = 0;
    std::vector<float> a(px), b(px);
    std::vector<int> ca(px), cb(px);
    for (int s = 0; s < n_seq; ++s) {
      batch[s].downloadDepthmap(&a[0]); alone[s]->downloadDepthmap(&b[0]);
      bad += memcmp(&a[0], &b[0], px * sizeof(float)) != 0;
      batch[s].downloadConvergence(&ca[0]); alone[s]->downloadConvergence(&cb[0]);
      bad += memcmp(&ca[0], &cb[0], px * sizeof(int)) != 0;
      bad += batch[s].getConvergedCount() != alone[s]->getConvergedCount();
      rmd::DepthmapDenoiser den(w, h);
      den.setLargeSigmaSq(range[1] - range[0]);
      den.denoise(batch[s].getMu(), batch[s].getSigmaSq(), batch[s].getA(), batch[s].getB(), &a[0], 0.5f, 20);
      den.denoise(alone[s]->getMu(), alone[s]->getSigmaSq(), alone[s]->getA(), alone[s]->getB(), &b[0], 0.5f, 20);
      bad += memcmp(&a[0], &b[0], px * sizeof(float)) != 0;
      printf("member %d: converged %zu\n", s, batch[s].getConvergedCount());
    }
    for (int s = 0; s < n_seq; ++s) delete alone[s];
    printf("%s\n", bad ? "MISMATCH" : "batch == stand-alone");
    return bad ? 1 : 0;
  } catch (const std::exception& e) {
    printf("caught: %s\n", e.what());
    return 3;
  }
}
